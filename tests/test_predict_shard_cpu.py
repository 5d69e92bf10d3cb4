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
    # one rasterizer call per frame in which this rank owns a camera, 2 views (colour + mask) per owned camera
    assert calls == [2 * len(shard.cams_of_frame(f)) for f in range(FRAMES) if shard.cams_of_frame(f)]
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
@pytest.mark.parametrize("world", [2, 3])
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
