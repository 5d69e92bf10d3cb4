"""Forward-only (frame, camera) sharding of predict.py's render loop (gsdyn/predict.py) on CPU:
gloo, world_size 2 and 3, oracle-backed TEST DOUBLE as the rasterizer.  The union of the ranks' images equals the single-rank
list image for image, every pair is rendered exactly once, and a rank renders its cameras of a frame in one call."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
P, W, H, CAMS, FRAMES = 90, 40, 24, 4, 3


def _setup():
    for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _install_double(calls=None):
    import oracle_double
    from diff_gaussian_rasterization import _hip

    def fwd_batch(*a, **k):
        if calls is not None:
            calls.append(len(a[0]))
        return oracle_double.rasterize_forward_batch(*a, **k)
    _hip.rasterize_forward = oracle_double.rasterize_forward
    _hip.rasterize_forward_batch = fwd_batch
    _hip.final_transmittance = oracle_double.final_transmittance


def _scene():
    from gsdyn import params2rendervar, synth_scene_params
    params = synth_scene_params(P, device="cpu", scale_lo=0.05, scale_hi=0.25)
    frames = []
    with torch.no_grad():
        for f in range(FRAMES):     # the "rollout": the Gaussians drift a little from frame to frame
            d = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
            d["means3D"] = d["means3D"] + 0.02 * f
            frames.append(d)
    return frames


def _worker(rank, world, port, out_dir):
    _setup()
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    _install_double(calls)
    from gsdyn.predict import FrameShard, gather_frames, ring_poses, shard_pairs
    shard = FrameShard("cpu", W, H, ring_poses(CAMS, W, H))
    assert shard.rank == rank and shard.world == world
    local = shard.render_episode(_scene())
    assert sorted(local) == sorted(shard_pairs(FRAMES, CAMS, rank, world))
    # one rasterizer call per frame in which this rank owns a camera, ONE view per owned camera (the mask comes from its final_T)
    assert calls == [len(shard.cams_of_frame(f)) for f in range(FRAMES) if shard.cams_of_frame(f)]
    merged = gather_frames(local)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), **{f"{f}_{c}_{i}": t.numpy() for (f, c), v in merged.items() for i, t in enumerate(v)})
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_pairs_partition():
    _setup()
    from gsdyn.predict import shard_pairs
    for world in (1, 2, 3, 8):
        got = sorted(p for r in range(world) for p in shard_pairs(5, 4, r, world))
        assert got == [(f, c) for f in range(5) for c in range(4)]
    assert shard_pairs(2, 4, 3, 8) == [(0, 3)] and shard_pairs(2, 4, 5, 8) == [(1, 1)]   # 8 GPUs: one camera, every second frame


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_renders_equal_single_rank(tmp_path, world):
    _setup()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(tmp_path / "merged.npz")
    _install_double()
    from gsdyn.predict import FrameShard, ring_poses
    ref = FrameShard("cpu", W, H, ring_poses(CAMS, W, H), rank=0, world=1).render_episode(_scene())
    assert len(ref) == FRAMES * CAMS and len(z.files) == 3 * FRAMES * CAMS
    for (f, c), v in ref.items():
        for i, t in enumerate(v):
            assert np.array_equal(z[f"{f}_{c}_{i}"], t.numpy()), (f, c, i)
        assert float(v[2].max()) <= 1.0 + 1e-5 and float(v[2].max()) > 0.1      # the mask render: accumulated alpha


# ------------------------------------------------------------------------------------------ predict_episode: rollout + sharded renders as one call
EP_P, EP_STEPS = 260, 5
ROLL = dict(max_nobj=12, fps_radius=0.12, adj_thresh=0.35, topk=4, connect_all=False, dist_thresh=0.004, n_fps_all=60, remove_outliers=True)


def _episode_inputs():
    from gsdyn import synth_scene_params
    from gsdyn.dynamics import DynamicsPredictor
    cfg = dict(nf_particle=16, nf_relation=16, nf_effect=16, attr_dim=2, state_dim=0, action_dim=3, pstep=2, rel_attr_dim=2,
               rel_group_dim=1, rel_distance_dim=3, n_his=3)
    torch.manual_seed(0)
    model = DynamicsPredictor(cfg).eval()
    params = {k: v.detach() for k, v in synth_scene_params(EP_P, device="cpu", scale_lo=0.05, scale_hi=0.2).items()}
    params["means3D"] = params["means3D"] * 0.4
    eef = torch.tensor([[0.5, 0.1, 0.0]]) + torch.tensor([[0.03, 0.0, 0.01]]) * torch.tensor([0.0, 1.0, 1.05, 2.0, 3.0])[:, None]   # step 2 moves < dist_thresh? no: 0.0016
    return model, params, eef


def _episode_worker(rank, world, port, out_dir):
    _setup()
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_double()
    from gsdyn.predict import predict_episode, ring_poses
    model, params, eef = _episode_inputs()
    frames, vis, tm = predict_episode(model, params, eef, ring_poses(CAMS, W, H), W, H, rollout_cfg=ROLL, gather_to=0, rgba=True)
    assert tm["frames"] == EP_STEPS and len(vis) == EP_STEPS
    if rank == 0:
        np.savez(os.path.join(out_dir, "episode.npz"), **{f"{f}_{c}_{i}": t.numpy() for (f, c), v in frames.items() for i, t in enumerate(v)})
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_predict_episode_is_rollout_then_sharded_renders(tmp_path):
    """predict_episode on 2 ranks == collect_scene_data (rollout -> smoothing -> packing) followed by the per-piece renders on one:
    the composition of /root/reference/src/predict.py:74-164 (one call per rank, every rank rolls out, pairs dealt round-robin)."""
    _setup()
    mp.spawn(_episode_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "episode.npz")
    _install_double()
    from gsdyn.dynamics import pack_scene_data, remove_statistical_outliers, rollout, smooth_frames
    from gsdyn.predict import FrameShard, collect_scene_data, compose_rgba, ring_poses
    model, params, eef = _episode_inputs()
    scene_sorted, vis, tm = collect_scene_data(model, params, eef, **ROLL)          # as predict_episode runs it: Morton order
    scene, _, _ = collect_scene_data(model, params, eef, spatial_sort=False, **ROLL)
    from gsdyn.dynamics import spatial_order
    perm = spatial_order(scene[0]["means3D"])
    assert sorted(perm.tolist()) == list(range(scene[0]["means3D"].shape[0])) and not torch.equal(perm, torch.arange(perm.numel()))
    for t, (a, b) in enumerate(zip(scene_sorted, scene)):    # one permutation for the whole episode, every per-Gaussian array.  The order is
        for k in a:                                          # applied BEFORE the rollout (the inlier list mapped through its inverse: the same
            assert torch.equal(a[k], b[k][perm]), (t, k)     # farthest-point picks); the skinning is per Gaussian, on the host as on the device
    # streaming mode (frames handed over while the rollout goes on: predict_episode(overlap=True)) == batch mode, bit for bit, in order
    got = []
    scene_stream, vis_stream, _ = collect_scene_data(model, params, eef, on_frame=lambda t, d, ev: got.append((t, d, ev)), **ROLL)
    assert [t for t, _, _ in got] == list(range(EP_STEPS)) and all(ev is None for _, _, ev in got) and len(scene_stream) == EP_STEPS
    for (t, d, _), b in zip(got, scene_sorted):
        assert d is scene_stream[t]
        for k in b:
            assert torch.equal(d[k], b[k]), (t, k)
    for a, b in zip(vis_stream, vis):
        assert np.array_equal(a["kp"], b["kp"]) and np.array_equal(a["tool_kp"], b["tool_kp"])
    # the pieces, called one by one as the reference's collect_scene_data strings them together
    op = torch.sigmoid(params["logit_opacities"])
    keep = op[:, 0] >= 0.1
    xyz0, rgb0 = params["means3D"][keep], params["rgb_colors"][keep]
    q0 = torch.nn.functional.normalize(params["unnorm_rotations"])[keep]
    inl = remove_statistical_outliers(xyz0)
    out = rollout(model, xyz0, rgb0, q0, op[keep], eef[:, None, :], EP_STEPS, inl, max_nobj=ROLL["max_nobj"], fps_radius_value=ROLL["fps_radius"],
                  adj_thresh=ROLL["adj_thresh"], topk=ROLL["topk"], connect_all=False, dist_thresh=ROLL["dist_thresh"], n_fps_all=ROLL["n_fps_all"])
    out = smooth_frames(*out)
    scene2, _ = pack_scene_data(out[0], out[1], out[2], out[3], torch.exp(params["log_scales"])[keep], out[4], out[5])
    assert len(scene) == len(scene2) == EP_STEPS
    for a, b in zip(scene, scene2):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert float((scene[-1]["means3D"] - scene[0]["means3D"]).abs().max()) > 1e-4      # the rollout moved the Gaussians
    ref = FrameShard("cpu", W, H, ring_poses(CAMS, W, H), rank=0, world=1).render_episode(scene_sorted)
    assert len(z.files) == 3 * EP_STEPS * CAMS
    for (f, c), (im, depth, mask) in ref.items():
        assert np.array_equal(z[f"{f}_{c}_0"], compose_rgba(im, mask).numpy()), (f, c)
        assert np.array_equal(z[f"{f}_{c}_1"], depth.numpy()) and np.array_equal(z[f"{f}_{c}_2"], mask.numpy())


# ------------------------------------------------------------------------------------------ the pipelined episode: one rank rolls out, the others skin + render
def _pipelined_worker(rank, world, port, out_dir, producer_renders, light, fault=None):
    _setup()
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_double()
    from gsdyn import dynamics as D
    from gsdyn.predict import predict_episode, render_ranks_of, ring_poses, shard_pairs
    model, params, eef = _episode_inputs()
    ran = {"fps": 0, "gnn": 0}
    fps0, fwd0 = D.farthest_point_sampler, D.DynamicsPredictor.forward
    D.farthest_point_sampler = lambda *a, **k: (ran.__setitem__("fps", ran["fps"] + 1), fps0(*a, **k))[1]
    D.DynamicsPredictor.forward = lambda self, *a, **k: (ran.__setitem__("gnn", ran["gnn"] + 1), fwd0(self, *a, **k))[1]
    scene = []
    light = light and rank == 0        # the producer asks for no scene back: it rolls out its tracked particles only
    if fault and rank == 0:
        # what a HIP producer's graphed rollout does when a bone needs the host's SVD: every packet has left, THEN it finds out
        # ("redo"); and a rollout that dies half way ("die": after its third packet)
        real = D.rollout

        def faulty(*a, graph_step=True, on_skin=None, **k):
            if fault == "die":
                sent = [0]

                def counting(i, pk):
                    on_skin(i, pk)
                    sent[0] += 1
                    if sent[0] == 3:
                        raise ValueError("injected rollout failure")
                return real(*a, graph_step=graph_step, on_skin=counting, **k)
            out = real(*a, graph_step=graph_step, on_skin=on_skin, **k)
            if graph_step:
                raise D.RolloutNeedsHostSVD("injected")
            return out
        D.rollout = faulty
    try:
        frames, vis, tm = predict_episode(model if rank == 0 else None, params, eef, ring_poses(CAMS, W, H), W, H, rollout_cfg=ROLL, gather_to=0,
                                          rgba=True, pipeline=True, producer_renders=producer_renders, scene_out=None if light else scene)
        assert fault != "die", "every rank must raise when the producer's rollout fails"
    except (ValueError, RuntimeError) as e:
        assert fault == "die" and ("injected" in str(e) if rank == 0 else "rollout on rank 0 failed" in str(e)), (rank, repr(e))
        dist.barrier()
        dist.destroy_process_group()
        return
    assert tm["rollout_attempts"] == (2 if fault == "redo" else 1)
    rr = render_ranks_of(world, 0, producer_renders)
    assert tm["pipelined"] and tm["render_ranks"] == rr and tm["frames"] == EP_STEPS and len(vis) == EP_STEPS
    assert len(scene) == (0 if light else EP_STEPS) and (rank != 0 or tm["producer_tracked_only"] == light)
    assert (tm["gaussians"] == ROLL["n_fps_all"]) == light
    # only the producer samples and runs the network; a render rank renders exactly its share of the pairs
    assert (ran["fps"] > 0 and ran["gnn"] > 0) if rank == 0 else (ran["fps"] == 0 and ran["gnn"] == 0)
    assert tm["pairs_on_this_rank"] == (len(shard_pairs(EP_STEPS, CAMS, rr.index(rank), len(rr))) if rank in rr else 0)
    np.savez(os.path.join(out_dir, f"scene_{rank}.npz"), **{f"{t}_{k}": v.numpy() for t, d in enumerate(scene) for k, v in d.items()},
             **{f"kp_{t}": v["kp"] for t, v in enumerate(vis)}, **{f"tool_{t}": v["tool_kp"] for t, v in enumerate(vis)})
    if rank == 0:
        np.savez(os.path.join(out_dir, "episode.npz"), **{f"{f}_{c}_{i}": t.numpy() for (f, c), v in frames.items() for i, t in enumerate(v)})
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,producer_renders,light", [(2, False, False), (3, False, True), (3, True, False)])
def test_pipelined_episode_equals_the_replicated_one(tmp_path, world, producer_renders, light):
    """predict_episode(pipeline=True): rank 0 rolls out and broadcasts one skinning packet per moving step; the other ranks never sample,
    never run the network -- they move the Gaussians with the packets and render.  Every rank ends up with the SAME per-frame render
    inputs and keypoints as the single-process episode, bit for bit, and the union of the render ranks' images is its images.
    ``light``: the producer asks for no scene back and renders nothing, so it rolls out its tracked particles ONLY -- the packets it
    sends, hence everybody's frames, and its keypoints are still the single-process episode's, bit for bit."""
    _setup()
    mp.spawn(_pipelined_worker, args=(world, _free_port(), str(tmp_path), producer_renders, light), nprocs=world, join=True)
    _install_double()
    from gsdyn.predict import collect_scene_data, compose_rgba, FrameShard, ring_poses
    model, params, eef = _episode_inputs()
    scene, vis, _ = collect_scene_data(model, params, eef, **ROLL)
    assert sum(int(not np.array_equal(vis[t]["tool_kp"], vis[t - 1]["tool_kp"])) for t in range(1, EP_STEPS)) >= 3   # frames that moved
    for r in range(world):
        z = np.load(tmp_path / f"scene_{r}.npz")
        for t, d in enumerate(scene):
            for k, v in d.items():
                if light and r == 0:
                    assert f"{t}_{k}" not in z.files
                else:
                    assert np.array_equal(z[f"{t}_{k}"], v.numpy()), (r, t, k)
            assert np.array_equal(z[f"kp_{t}"], vis[t]["kp"]) and np.array_equal(z[f"tool_{t}"], vis[t]["tool_kp"]), (r, t)
    ref = FrameShard("cpu", W, H, ring_poses(CAMS, W, H), rank=0, world=1).render_episode(scene)
    z = np.load(tmp_path / "episode.npz")
    assert len(z.files) == 3 * EP_STEPS * CAMS
    for (f, c), (im, depth, mask) in ref.items():
        assert np.array_equal(z[f"{f}_{c}_0"], compose_rgba(im, mask).numpy()), (f, c)
        assert np.array_equal(z[f"{f}_{c}_1"], depth.numpy()) and np.array_equal(z[f"{f}_{c}_2"], mask.numpy())


@pytest.mark.timeout(600)
def test_pipelined_episode_redoes_a_rollout_that_needs_the_host(tmp_path):
    """a graphed rollout finds out AFTER its last packet that a bone needs the host's SVD.  The producer must not broadcast
    the eager redo's packets into ranks that have consumed theirs (a hang): it ends the attempt with a status word, every rank drops its
    frames, and the episode runs again eagerly -- the same frames as ever."""
    _setup()
    mp.spawn(_pipelined_worker, args=(3, _free_port(), str(tmp_path), False, False, "redo"), nprocs=3, join=True)
    _install_double()
    from gsdyn.predict import collect_scene_data, compose_rgba, FrameShard, ring_poses
    model, params, eef = _episode_inputs()
    scene, vis, _ = collect_scene_data(model, params, eef, **ROLL)
    for r in range(3):
        z = np.load(tmp_path / f"scene_{r}.npz")
        for t, d in enumerate(scene):
            for k, v in d.items():
                assert np.array_equal(z[f"{t}_{k}"], v.numpy()), (r, t, k)
    ref = FrameShard("cpu", W, H, ring_poses(CAMS, W, H), rank=0, world=1).render_episode(scene)
    z = np.load(tmp_path / "episode.npz")
    assert len(z.files) == 3 * EP_STEPS * CAMS
    for (f, c), (im, depth, mask) in ref.items():
        assert np.array_equal(z[f"{f}_{c}_0"], compose_rgba(im, mask).numpy()), (f, c)


@pytest.mark.timeout(600)
def test_pipelined_episode_fails_on_every_rank_when_the_rollout_dies(tmp_path):
    """... and a producer whose rollout raises half way keeps the broadcasts matched (empty packets for the steps it did not reach) and
    tells the others: they raise instead of waiting for a packet that never comes."""
    _setup()
    mp.spawn(_pipelined_worker, args=(2, _free_port(), str(tmp_path), False, False, "die"), nprocs=2, join=True)


def test_skin_packet_round_trip():
    _setup()
    from gsdyn import dynamics as D
    g = torch.Generator().manual_seed(3)
    nb, cap = 5, 8
    bones, mot, pred = torch.randn(nb, 3, generator=g), torch.randn(nb, 3, generator=g), torch.randn(nb, 3, generator=g)
    rel = (torch.rand(nb, nb, generator=g) < 0.6).long()
    rel = ((rel + rel.T) > 0).long().fill_diagonal_(0)
    R, q = D.bone_transforms(bones, mot, rel)
    pk = D.pack_skin(cap, bones, R, mot, q, pred)
    assert pk.shape == (D.skin_packet_len(cap),) and float(pk[0]) == nb and float(pk[1]) == 1.0
    for a, b in zip(D.unpack_skin(pk, cap, n_valid=nb), (bones, R, mot, q, pred)):
        assert torch.equal(a, b)
    assert all(t.shape[0] == cap for t in D.unpack_skin(pk, cap)) and float(D.unpack_skin(pk, cap)[0][nb:].abs().max()) == 0.0
    xyz, quat = torch.randn(40, 3, generator=g), torch.nn.functional.normalize(torch.randn(40, 4, generator=g), dim=-1)
    a = D.interpolate_motions(bones, mot, rel, xyz, quat=quat)
    b = D.blend_skinning(*D.unpack_skin(pk, cap, n_valid=nb)[:4], xyz, quat)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_depth_cuts_policy_on_host_tensors():
    """gsdyn.render.DepthCuts without a device: the ping-pong of the proposal buffers, the dilation (a tile bins with the deepest proposal of
    its neighbourhood; +inf -- "my list ran out" -- spreads to the neighbours), the redo bookkeeping and the adaptation of the dilation."""
    _setup()
    from gsdyn.render import DepthCuts
    INF = 0x7f800000
    h, w = 48, 80                     # 3 x 5 tiles
    T = 15
    dc = DepthCuts(dilate=1, adapt=True)
    cin, cout, redo, margin = dc.arm("cams", 2, h, w, "cpu", 0)
    assert cin is None and len(cout) == 2 and cout[0].numel() == T and redo.tolist() == [0, 0] and margin == 1.01
    prop = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0])
    cout[0].copy_(prop.view(torch.int32)); cout[1].copy_((prop * 2).view(torch.int32))          # noqa: E702  (what the blend would write)
    cout[1][7] = INF
    dc.sent(redo)
    cin2, cout2, redo2, _ = dc.arm("cams", 2, h, w, "cpu", 1)
    assert cout2[0].data_ptr() != cout[0].data_ptr()                  # the other buffer of the pair
    want0 = torch.nn.functional.max_pool2d(prop.view(1, 1, 3, 5), 3, stride=1, padding=1).reshape(-1)
    assert torch.equal(cin2[0].view(torch.float32), want0)
    got1 = cin2[1].view(3, 5)
    assert (got1[0:3, 1:4] == INF).all() and int(got1[0, 0]) != INF and float(got1[0, 0].view(torch.float32)) == 14.0   # max of (1,2,6,7) x 2
    redo2[1] = 3                                                      # the blend found three failing tiles in view 1
    dc.sent(redo2)
    cin3, cout3, redo3, _ = dc.arm("cams", 2, h, w, "cpu", 2)
    assert cout3[0].data_ptr() == cout[0].data_ptr() and dc.dilate == 2          # adapted: the earlier frame's words were there to read
    dc.sent(redo3)
    assert dc.failed() == {1: [1]} and dc.failed() == {} and dc.calls == 3 and dc.cut_calls == 2 and dc.redone == 1
    for f in range(16):                                               # sixteen clean frames: one tile narrower again
        a = dc.arm("cams", 2, h, w, "cpu", 3 + f)
        dc.sent(a[2])
    dc.arm("cams", 2, h, w, "cpu", 99)
    assert dc.dilate == 1
    other = dc.arm("other cameras", 1, h, w, "cpu", 100)
    assert other[0] is None                                           # a new camera set starts without cuts
    assert DepthCuts().arm("k", 1, 2160, 3840, "cpu", 0) is None      # 135 x 240 tiles: not served
    dc.dilate = 5
    dc.failed()
    assert DepthCuts().dilate == 5 and DepthCuts(dilate=3).dilate == 3                      # the next episode of the process starts where this one ended
    DepthCuts._learned["dilate"] = 2
