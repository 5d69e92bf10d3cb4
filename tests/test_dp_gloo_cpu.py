"""Multi-process test of the view-sharded data-parallel step on CPU (gloo; world_size 2 with 4 views, 4 with 8 views, 8 with
12 views -- more views than ranks, uneven shards -- and 8 with 8 views, the one-view-per-GPU layout of BASELINE configs[3]):
the all-reduced flat gradient bucket equals the sum of single-process per-view gradients, replicas take
the identical Adam step, and densification statistics are reduced.
The rasterizer backend is replaced by the oracle-backed TEST DOUBLE (no GPU here); the DP driver,
bucket layout and collectives are the product code under test."""
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

P, W, H, V = 120, 48, 32, 4


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


def _install_double():
    import oracle_double
    from diff_gaussian_rasterization import _hip
    _hip.rasterize_forward = oracle_double.rasterize_forward
    _hip.rasterize_backward = oracle_double.rasterize_backward
    _hip.rasterize_forward_batch = oracle_double.rasterize_forward_batch
    _hip.rasterize_backward_batch = oracle_double.rasterize_backward_batch


def _make_problem(V=V):
    from gsdyn import synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.step import make_rigidity_variables
    params = synth_scene_params(P, device="cpu", scale_lo=0.05, scale_hi=0.25)
    cams = synth_ring_cameras(V, W, H, device="cpu")
    views = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=10 + i, device="cpu")
        views.append(dict(cam=cam, im=im, seg=seg, id=i))
    rig = make_rigidity_variables(params, num_knn=5)
    return params, views, rig


def _variables(rig):
    from gsdyn.dp import init_variables
    v = init_variables(P, "cpu")
    v.update({k: t.clone() for k, t in rig.items()})
    return v


def _worker(rank, world, port, out_dir, V=V):
    _setup_paths()
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_double()
    from gsdyn import LossWeights, initialize_optimizer
    from gsdyn.dp import ViewShardedStep, shard_views
    params, views, rig = _make_problem(V)
    opt = initialize_optimizer(params, scene_radius=4.0)
    stepper = ViewShardedStep(params, opt, LossWeights(), density_stats=True)   # exercise the statistics collectives at t > 0 too
    assert shard_views(V, rank, world) == list(range(rank, V, world))
    variables = _variables(rig)
    before = {k: p.detach().clone() for k, p in params.items()}
    total, variables = stepper(views, variables, is_initial_timestep=False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=stepper.bucket.flat.numpy(), loss=total.numpy(),
             accum=variables["means2D_gradient_accum"].numpy(), denom=variables["denom"].numpy(),
             maxrad=variables["max_2D_radius"].numpy(),
             **{"after_" + k: p.detach().numpy() for k, p in params.items()},
             **{"before_" + k: v.numpy() for k, v in before.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,V", [(2, 4), (4, 8), (8, 12), (8, 8)])
def test_view_sharded_step(tmp_path, world, V):
    _setup_paths()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), V), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0, r1 = ranks[0], ranks[1]
    # replicas agree bit-for-bit after the step (same reduced gradients, same Adam update)
    for rk in ranks[1:]:
        for k in r0.files:
            if k.startswith("after_") or k in ("flat", "accum", "denom", "maxrad"):
                assert np.array_equal(r0[k], rk[k]), k
    assert float(r0["loss"]) != float(r1["loss"])  # different shards

    # single-process reference: all V views on one rank, same code path minus the collective
    _install_double()
    from gsdyn import LossWeights, initialize_optimizer
    from gsdyn.dp import ViewShardedStep
    params, views, rig = _make_problem(V)
    opt = initialize_optimizer(params, scene_radius=4.0)
    stepper = ViewShardedStep(params, opt, LossWeights(), density_stats=True)
    variables = _variables(rig)
    total, variables = stepper(views, variables, is_initial_timestep=False)
    flat = stepper.bucket.pack().numpy()    # a single rank never packs by itself
    scale = np.abs(flat).max()
    assert np.abs(flat - r0["flat"]).max() <= 1e-5 * scale      # sum order differs: 1e-5 rel (section 8e)
    np.testing.assert_allclose(float(total), sum(float(rk["loss"]) for rk in ranks), rtol=1e-5)
    np.testing.assert_allclose(variables["denom"].numpy(), r0["denom"])
    np.testing.assert_allclose(variables["means2D_gradient_accum"].numpy(), r0["accum"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(variables["max_2D_radius"].numpy(), r0["maxrad"])
    moved = False
    for k, p in params.items():
        np.testing.assert_allclose(p.detach().numpy(), r0["after_" + k], rtol=1e-4, atol=1e-6)
        moved |= not np.array_equal(r0["after_" + k], r0["before_" + k])
    assert moved


def test_grad_bucket_layout():
    _setup_paths()
    from gsdyn import synth_scene_params
    from gsdyn.dp import GradBucket
    params = synth_scene_params(10, device="cpu")
    b = GradBucket(params)
    assert "rgb_colors" not in b.names                      # frozen in the reference (train_utils.py:133)
    assert b.flat.numel() == 10 * (3 + 3 + 4 + 1 + 3) + 2 * 50 * 3
    b.zero()
    assert all(p.grad is None for p in b.params)            # set-to-none: the next backward hands its tensors over
    params["means3D"].grad = torch.ones_like(params["means3D"])
    flat = b.pack()
    s, e = b.slices["means3D"]
    assert torch.all(flat[s:e] == 1.0) and float(flat.sum()) == 30.0   # missing gradients packed as zeros
    assert params["means3D"].grad.data_ptr() == flat[s:e].data_ptr()  # after packing, .grad IS the bucket slice
    params["means3D"].grad.add_(1.0)
    assert torch.all(b.flat[s:e] == 2.0)


def test_grad_bucket_in_place_producers_skip_the_pack_copy(monkeypatch):
    """A producer that wrote its gradient into ``views()`` (the direct step's backward): ``pack`` finds it in place -- no ``torch.cat`` --,
    copies a gradient that arrived as a tensor of its own, and clears the slice of a parameter without a gradient only when the bucket
    was written since that slice was last known to be zero."""
    _setup_paths()
    from gsdyn import synth_scene_params
    from gsdyn.dp import GradBucket
    params = synth_scene_params(10, device="cpu")
    b = GradBucket(params)
    views = b.views()
    assert set(views) == set(b.names) and all(views[k].shape == params[k].shape for k in b.names)
    calls = {"cat": 0, "zero": 0}
    real_cat, real_zero = torch.cat, torch.Tensor.zero_
    monkeypatch.setattr(torch, "cat", lambda *a, **k: (calls.__setitem__("cat", calls["cat"] + 1), real_cat(*a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "zero_", lambda self: (calls.__setitem__("zero", calls["zero"] + 1), real_zero(self))[1])
    for step in range(3):
        b.zero()
        views["means3D"].data.fill_(float(step + 1))            # "kernel" writes through the raw pointer ...
        params["means3D"].grad = views["means3D"]                # ... and the step hands the slice over as .grad
        params["log_scales"].grad = torch.full_like(params["log_scales"], 2.0)       # a tensor of its own (autograd)
        flat = b.pack()
        s, e = b.slices["means3D"]
        assert torch.all(flat[s:e] == step + 1)
        s, e = b.slices["log_scales"]
        assert torch.all(flat[s:e] == 2.0) and params["log_scales"].grad.data_ptr() == flat[s:e].data_ptr()
        for k in b.names:
            if k not in ("means3D", "log_scales"):
                s, e = b.slices[k]
                assert torch.all(flat[s:e] == 0.0), k
    assert calls["cat"] == 0 and calls["zero"] == 0      # steady state: no copy of the in-place gradients, no clearing of the known-zero slices
    # a user who scribbles into a slice that is later missing: it is cleared again
    b.zero()
    views["seg_colors"].add_(5.0)
    params["means3D"].grad = views["means3D"]
    flat = b.pack()
    s, e = b.slices["seg_colors"]
    assert torch.all(flat[s:e] == 0.0)
    # nothing in place: the one-cat path, as before
    b.zero()
    params["means3D"].grad = torch.ones_like(params["means3D"])
    n = calls["cat"]
    flat = b.pack()
    assert calls["cat"] == n + 1 and float(flat.sum()) == 30.0
