"""The multi-rank code on the REAL kernels, on ONE GPU: ``mp.spawn`` starts 2 / 4 ranks that all use ``cuda:0``
with the HIP rasterizer (no test double) and a gloo process group whose all-reduce goes through host copies (what
``bench.py``'s ``GSR_BENCH_SINGLE_DEVICE=1`` does).  Checked:

  * ``ViewShardedStep`` (gsdyn/dp.py): the reduced flat bucket equals the sum of the per-view HIP gradients
    (one single-process call per view, summed in fp64) to 1e-5 of its maximum, the backward wrote its gradients INTO the bucket
    (no packing copy), and the replicas are bit-identical after the Adam step;
  * the bench step (``render_step_views(grad_out=bucket.views())`` + ``GradBucket.all_reduce`` + FusedAdam): the same identities;
  * ``FrameShard`` (gsdyn/predict.py; row E2): the union of the ranks' (frame, camera) renders equals the single-process list bit for
    bit, every pair rendered exactly once.

What this cannot show is RCCL itself (one GPU: the collective is gloo on host copies) -- that is the driver's 8-GPU run.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
P, W, H = 20_000, 320, 240
CAMS, FRAMES = 4, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _host_collectives():
    """gloo on host copies of the device tensors (all ranks share cuda:0; NOT a performance path)."""
    real = dist.all_reduce

    def all_reduce(t, *a, **k):
        if t.is_cuda:
            h = t.cpu()
            real(h, *a, **k)
            t.copy_(h)
            return None
        return real(t, *a, **k)
    dist.all_reduce = all_reduce


def _init(rank, world, port):
    _setup_paths()
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _host_collectives()
    return torch.device("cuda", 0)


def _problem(dev, V):
    from gsdyn import synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    from gsdyn.step import make_rigidity_variables
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(V, W, H, device=dev)
    views = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=10 + i, device=dev)
        views.append(dict(cam=cam, im=im, seg=seg, id=i))
    variables = init_variables(P, dev)
    variables.update(make_rigidity_variables(params, num_knn=8))
    return params, views, variables


# ------------------------------------------------------------------------------------------ ViewShardedStep
def _dp_worker(rank, world, port, out_dir, V):
    dev = _init(rank, world, port)
    from gsdyn import LossWeights, initialize_optimizer
    from gsdyn.dp import ViewShardedStep
    params, views, variables = _problem(dev, V)
    opt = initialize_optimizer(params, scene_radius=4.0)            # FusedAdam; both colour groups have lr 0 -> the direct step
    stepper = ViewShardedStep(params, opt, LossWeights())
    assert stepper.frozen_colours and stepper.world == world and stepper.rank == rank
    import gsdyn.dp as dp_mod
    cats = []
    real_cat = torch.cat
    dp_mod.torch.cat = lambda *a, **k: (cats.append(1) if k.get("out") is stepper.bucket.flat else None, real_cat(*a, **k))[1]   # packing copies only
    try:
        for _ in range(2):        # the second step runs in capacity mode (no host wait inside the forward): the steady state
            before = {k: p.detach().clone() for k, p in params.items()}
            total, variables = stepper(views, variables, is_initial_timestep=False)
    finally:
        dp_mod.torch.cat = real_cat
    torch.cuda.synchronize()
    s, e = stepper.bucket.slices["means3D"]
    in_place = params["means3D"].grad.data_ptr() == stepper.bucket.flat[s:e].data_ptr()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=stepper.bucket.flat.cpu().numpy(), loss=total.cpu().numpy(),
             in_place=in_place, cats=len(cats),
             **{"after_" + k: p.detach().cpu().numpy() for k, p in params.items()},
             **{"before_" + k: v.cpu().numpy() for k, v in before.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,V", [(2, 4), (4, 4)])
def test_view_sharded_step_on_the_real_kernels(tmp_path, world, V):
    _setup_paths()
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path), V), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    for rk in ranks[1:]:          # replicas: same reduced bucket, same Adam update, bit for bit
        for k in r0.files:
            if k.startswith("after_") or k == "flat":
                assert np.array_equal(r0[k], rk[k]), k
    assert all(bool(rk["in_place"]) for rk in ranks)            # .grad IS the bucket slice
    assert all(int(rk["cats"]) == 0 for rk in ranks)            # ... and nothing was packed with torch.cat on the way
    assert any(not np.array_equal(r0["after_" + k], r0["before_" + k]) for k in ("means3D", "log_scales"))

    # single process, the REAL kernels again: the second step of the same two-step sequence, one call per VIEW, summed in fp64
    dev = torch.device("cuda", 0)
    from gsdyn import LossWeights, initialize_optimizer
    from gsdyn.dp import GradBucket, ViewShardedStep
    params, views, variables = _problem(dev, V)
    opt = initialize_optimizer(params, scene_radius=4.0)
    stepper = ViewShardedStep(params, opt, LossWeights())
    total, variables = stepper(views, variables, is_initial_timestep=False)          # step 1 (all views on one rank), Adam applied
    with torch.no_grad():            # continue from the replicas' exact parameters (Adam's first update is +-lr: a gradient near zero may take
        #                              the other sign under another summation order, so the parameters themselves are not compared here)
        for k, p in params.items():
            p.copy_(torch.tensor(r0["before_" + k], device=dev))
    acc = np.zeros(r0["flat"].shape, np.float64)
    losses = 0.0
    for d in views:
        one = ViewShardedStep(params, None, LossWeights())
        one.frozen_colours = True
        t, _ = one([d], dict(variables), is_initial_timestep=False, local_only=True)
        acc += one.bucket.pack().cpu().numpy().astype(np.float64)
        losses += float(t)
    scale = np.abs(acc).max()
    assert scale > 0
    assert np.abs(acc - r0["flat"]).max() <= 1e-5 * scale, np.abs(acc - r0["flat"]).max() / scale
    np.testing.assert_allclose(sum(float(rk["loss"]) for rk in ranks), losses, rtol=1e-5)


# ------------------------------------------------------------------------------------------ the bench step
def _bench_worker(rank, world, port, out_dir, V):
    dev = _init(rank, world, port)
    from gsdyn import initialize_optimizer, synth_ring_cameras, synth_scene_params
    from gsdyn.dp import GradBucket, shard_views
    from gsdyn.step import render_step_views
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.01, scale_hi=0.06)
    params["rgb_colors"].requires_grad_(True)
    cams_all = synth_ring_cameras(V, W, H, device=dev)
    dL_all = torch.tensor(np.random.default_rng(1234).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    ids = shard_views(V, rank, world)
    cams, dL = [cams_all[i] for i in ids], dL_all[ids].contiguous()
    bucket = GradBucket(params)
    opt = initialize_optimizer(params, 4.0)
    keys = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")
    placed = []
    for _ in range(3):                   # step 1 sizes the buffers synchronously; 2 and 3 are the steady state (fused activations, capacity mode)
        before = {k: p.detach().clone() for k, p in params.items()}
        bucket.zero()
        _, g = render_step_views(params, cams, dL, want_colour_grad=True, grad_out=bucket.views())
        for k in keys:
            params[k].grad = g.get(k)
        placed.append(all(g[k].data_ptr() == bucket.views()[k].data_ptr() for k in keys))
        bucket.all_reduce()
        opt.step()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=bucket.flat.cpu().numpy(), placed=np.array(placed),
             **{"after_" + k: p.detach().cpu().numpy() for k, p in params.items()},
             **{"before_" + k: v.cpu().numpy() for k, v in before.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,V", [(2, 8), (4, 8)])
def test_bench_step_sharded_on_the_real_kernels(tmp_path, world, V):
    """bench.py's N > 1 step: this rank's views through ``render_step_views`` with the gradients written straight into the bucket,
    one all-reduce, FusedAdam -- the reduced bucket = sum over ALL views of the single-view HIP gradients, replicas bit-identical."""
    _setup_paths()
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), V), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    for rk in ranks[1:]:
        for k in r0.files:
            if k.startswith("after_") or k == "flat":
                assert np.array_equal(r0[k], rk[k]), k
        assert rk["placed"].tolist() == r0["placed"].tolist()
    assert r0["placed"].tolist()[1:] == [True, True], r0["placed"]      # steady state: every gradient produced in place
    dev = torch.device("cuda", 0)
    from gsdyn import synth_ring_cameras, synth_scene_params
    from gsdyn.dp import GradBucket
    from gsdyn.step import render_step_views
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.01, scale_hi=0.06)
    params["rgb_colors"].requires_grad_(True)
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(torch.tensor(r0["before_" + k], device=dev))
    cams_all = synth_ring_cameras(V, W, H, device=dev)
    dL_all = torch.tensor(np.random.default_rng(1234).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    bucket = GradBucket(params)
    acc = np.zeros(r0["flat"].shape, np.float64)
    for v in range(V):                                       # one view per call, fresh tensors, summed in fp64
        bucket.zero()
        _, g = render_step_views(params, [cams_all[v]], dL_all[v:v + 1].contiguous(), want_colour_grad=True)
        for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"):
            params[k].grad = g.get(k)
        acc += bucket.pack().cpu().numpy().astype(np.float64)
    scale = np.abs(acc).max()
    assert scale > 0 and np.abs(acc - r0["flat"]).max() <= 1e-5 * scale, np.abs(acc - r0["flat"]).max() / scale


# ------------------------------------------------------------------------------------------ forward-only (frame, camera) sharding
def _scene(dev):
    from gsdyn import params2rendervar, synth_scene_params
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.01, scale_hi=0.06)
    frames = []
    with torch.no_grad():
        for f in range(FRAMES):       # the "rollout": the Gaussians drift a little from frame to frame
            d = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
            d["means3D"] = d["means3D"] + 0.02 * f
            frames.append(d)
    return frames


def _predict_worker(rank, world, port, out_dir):
    dev = _init(rank, world, port)
    from gsdyn.predict import FrameShard, ring_poses, shard_pairs
    shard = FrameShard(dev, W, H, ring_poses(CAMS, W, H))
    assert shard.rank == rank and shard.world == world
    local = shard.render_episode(_scene(dev))
    torch.cuda.synchronize()
    assert sorted(local) == sorted(shard_pairs(FRAMES, CAMS, rank, world))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=np.array(sorted(local)),
             **{f"im_{f}_{c}": v[0].cpu().numpy() for (f, c), v in local.items()},
             **{f"depth_{f}_{c}": v[1].cpu().numpy() for (f, c), v in local.items()},
             **{f"mask_{f}_{c}": v[2].cpu().numpy() for (f, c), v in local.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_frame_shards_on_the_real_kernels(tmp_path, world):
    _setup_paths()
    mp.spawn(_predict_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    from gsdyn.predict import FrameShard, ring_poses
    want = FrameShard(dev, W, H, ring_poses(CAMS, W, H), rank=0, world=1).render_episode(_scene(dev))
    torch.cuda.synchronize()
    seen = set()
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for f, c in z["keys"].tolist():
            assert (f, c) not in seen
            seen.add((f, c))
            for name, idx in (("im", 0), ("depth", 1), ("mask", 2)):
                assert np.array_equal(z[f"{name}_{f}_{c}"], want[(f, c)][idx].cpu().numpy()), (r, f, c, name)
    assert seen == set(want)


# ------------------------------------------------------------------------------------------ the pipelined episode (one rank rolls out, the others skin + render)
EP_P, EP_W, EP_H, EP_S = 30_000, 480, 272, 6
EP_ROLL = dict(max_nobj=100, fps_radius=0.3, adj_thresh=0.6, topk=5, connect_all=False, dist_thresh=0.005, n_fps_all=1000, remove_outliers=False)


def _episode_problem(dev):
    from gsdyn import synth_scene_params
    from gsdyn.dynamics import DynamicsPredictor
    gold = np.load(os.path.join(HERE, "golden", "dynamics_host.npz"))
    cfg = {str(k): int(v) for k, v in zip(gold["gnn_cfg_keys"], gold["gnn_cfg_vals"])}
    model = DynamicsPredictor(cfg, device=dev).eval()
    model.load_state_dict({k[len("gnn_w_"):]: torch.tensor(gold[k]) for k in gold.files if k.startswith("gnn_w_")})
    params = {k: v.detach() for k, v in synth_scene_params(EP_P, device=dev, scale_lo=0.01, scale_hi=0.04).items()}
    eef = torch.tensor([[0.04, 0.0, 0.02]], device=dev) * torch.tensor([0.0, 1.0, 1.01, 2.0, 3.0, 4.0], device=dev)[:, None]   # step 2: a repeated frame
    return model, params, eef


def _pipelined_worker(rank, world, port, out_dir, light):
    dev = _init(rank, world, port)
    from gsdyn.predict import predict_episode, render_ranks_of, ring_poses, shard_pairs
    model, params, eef = _episode_problem(dev)
    scene = []
    light = light and rank == 0        # a producer that wants no scene back rolls out its tracked particles only
    frames, vis, tm = predict_episode(model if rank == 0 else None, params, eef, ring_poses(CAMS, EP_W, EP_H), EP_W, EP_H, rollout_cfg=EP_ROLL,
                                      pipeline=True, scene_out=None if light else scene)
    torch.cuda.synchronize()
    rr = render_ranks_of(world)
    assert tm["pipelined"] and rr == list(range(1, world)) and len(scene) == (0 if light else EP_S) and len(vis) == EP_S
    assert rank != 0 or (tm["producer_tracked_only"] == light and (tm["gaussians"] == EP_ROLL["n_fps_all"]) == light)
    assert sorted(frames) == (sorted(shard_pairs(EP_S, CAMS, rr.index(rank), len(rr))) if rank in rr else [])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=np.array(sorted(frames)).reshape(-1, 2),
             **{f"sc_{t}_{k}": v.cpu().numpy() for t, d in enumerate(scene) for k, v in d.items() if k != "means2D"},
             **{f"kp_{t}": v["kp"] for t, v in enumerate(vis)},
             **{f"fr_{f}_{c}_{i}": t.cpu().numpy() for (f, c), v in frames.items() for i, t in enumerate(v)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,light", [(2, False), (3, True)])
def test_pipelined_episode_on_the_real_kernels(tmp_path, world, light):
    """predict_episode(pipeline=True) with 2 / 3 processes on this GPU (gloo: the packets travel as host tensors): rank 0 runs the graphed
    rollout and broadcasts the skinning packets, the others produce every frame from the packets with gsr_lbs alone -- the SAME render
    inputs and keypoints bit for bit (the rollout itself differs by ulps from run to run: every rank is compared with rank 0 of ITS run)
    -- and their renders are the single-process renders of those inputs, every (frame, camera) pair exactly once.  ``light``: rank 0
    asks for no scene back, so it rolls out its ~1000 tracked particles ONLY (collect_scene_data(tracked_only=True)): its keypoints are
    still the render ranks', bit for bit, and the render ranks' frames still move (the packets are the same kind)."""
    _setup_paths()
    mp.spawn(_pipelined_worker, args=(world, _free_port(), str(tmp_path), light), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    from gsdyn.predict import FrameShard, ring_poses
    zp = np.load(tmp_path / "rank0.npz")
    assert zp["keys"].size == 0                                  # the producer renders nothing by default
    assert light == (not any(n.startswith("sc_") for n in zp.files))
    z0 = np.load(tmp_path / "rank1.npz") if light else zp       # whose scene the others are compared with
    for t in range(EP_S):
        assert np.array_equal(zp[f"kp_{t}"], z0[f"kp_{t}"]), t
    scene = []
    for t in range(EP_S):
        d = {k: torch.tensor(z0[f"sc_{t}_{k}"], device=dev) for k in ("means3D", "colors_precomp", "rotations", "opacities", "scales")}
        d["means2D"] = torch.zeros_like(d["means3D"])
        scene.append(d)
    assert float((scene[-1]["means3D"] - scene[0]["means3D"]).norm(dim=-1).max()) > 1e-3
    mid = torch.lerp(scene[1]["means3D"], scene[3]["means3D"], 0.5)   # (the Morton permutation is per episode: rows correspond)
    assert float((scene[2]["means3D"] - mid).abs().max()) < 1e-6
    want = FrameShard(dev, EP_W, EP_H, ring_poses(CAMS, EP_W, EP_H), rank=0, world=1).render_episode(scene)
    torch.cuda.synchronize()
    seen = set()
    for r in range(1, world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for name in z0.files:
            if name.startswith(("sc_", "kp_")):
                assert np.array_equal(z[name], z0[name]), (r, name)
        for f, c in z["keys"].tolist():
            assert (f, c) not in seen
            seen.add((f, c))
            for i in range(3):
                assert np.array_equal(z[f"fr_{f}_{c}_{i}"], want[(f, c)][i].cpu().numpy()), (r, f, c, i)
    assert seen == set(want)


@pytest.mark.timeout(600)
def test_bench_reduce_leg_over_rccl_with_one_rank(tmp_path):
    """``bench.py`` with GSR_BENCH_FORCE_DIST=1: the ``nccl`` (= RCCL) process group on this box's one GPU, the step's reduce leg on the
    flat bucket through RCCL, the all-reduce timing pass and the ``rccl_ranks`` probe -- every RCCL call of the N > 1 path, over a
    communicator of one (RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    env = dict(os.environ)
    env.update(GSR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--views", "2", "--steps", "3", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])          # the JSON line is the LAST line of stdout (RCCL's banner is flushed before it)
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["allreduce_us"] is not None and 0.0 < line["allreduce_us"] < 5000.0
    assert line["value"] > 0 and line["config"]["views_total"] == 2
