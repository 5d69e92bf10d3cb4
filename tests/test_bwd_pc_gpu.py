"""The producer / consumer form of the backward blend (render_bwd_pc: round 5; the launcher's choice for dense scenes, GSR_BWD_PC=0/1
forces it off / on) must write exactly the records the barrier form writes: every gradient of a multi-view fwd + bwd bit for bit, all colour modes.  The launcher reads the switch once per
process, so each arm runs in its own interpreter (tools/r05_pc_check.py is the worker)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tools", "r05_pc_check.py")


def _grads(tmp_path, pc, V, P, S, frozen):
    out = str(tmp_path / f"g{pc}.npz")
    env = dict(os.environ, GSR_BWD_PC=str(pc), FROZEN=str(int(frozen)))
    subprocess.run([sys.executable, WORKER, out, str(V), str(P), str(S)], check=True, env=env, timeout=300,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("V,P,S,frozen", [(2, 20000, 400, False), (3, 5000, 200, True), (1, 30000, 304, False)])
def test_producer_consumer_backward_is_bit_identical(tmp_path, V, P, S, frozen):
    a = _grads(tmp_path, 0, V, P, S, frozen)
    b = _grads(tmp_path, 1, V, P, S, frozen)
    assert set(a.files) == set(b.files) and len(a.files) >= 5
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"{k}: producer / consumer backward differs from the barrier form"


@pytest.mark.gpu
def test_timed_out_wait_is_reported_by_the_next_backward(dev):
    """render_bwd_pc's waits are bounded; one that runs out marks a pinned host word and the launch's gradients are garbage.  The product
    must say so: the next backward launch fails with the reason (once -- the word is cleared), the one after works again."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(2000, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, 128, 96, device=dev)[0]

    def step():
        for v in params.values():
            v.grad = None
        im, _, _ = GaussianRasterizer(raster_settings=cam)(**params2rendervar(params))
        im.sum().backward()
        torch.cuda.synchronize()
        return params["means3D"].grad.clone()
    g0 = step()
    assert _hip.load_library().gsr_debug_pc_inject_error() == 0
    with pytest.raises(RuntimeError, match="render_bwd_pc"):
        step()
    assert torch.equal(step(), g0)


@pytest.mark.gpu
def test_aborted_blend_backward_poisons_its_own_call(dev):
    """the host only hears of a timed-out wait at its NEXT backward launch.  The gradients of the launch that aborted must not be
    consumed silently in between: the call's own error word (device memory, zeroed by its forward) makes the per-Gaussian backward queued
    behind the blend write NaN for dL/dmeans3D -- and only for that call."""
    import ctypes as C
    import torch
    from diff_gaussian_rasterization import _hip
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(2000, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, 128, 96, device=dev)[0]
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
    args = (rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"], rv["rotations"], None)
    dL = torch.ones((3, 96, 128), device=dev)

    def run(mark):
        _, radii, _, st = _hip.rasterize_forward(cam, *args)
        if mark:
            lib = _hip.load_library()
            lib.gsr_debug_pc_mark_call_error.restype = C.c_int
            lib.gsr_debug_pc_mark_call_error.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
            assert lib.gsr_debug_pc_mark_call_error(st.H, st.W, _hip._ptr(st.image), _hip._stream(dev)) == 0
        g = _hip.rasterize_backward(st, dL, rv["means3D"], radii, rv["colors_precomp"], None, rv["scales"], rv["rotations"], None)
        torch.cuda.synchronize()
        return g[0]
    g0 = run(False)
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    assert torch.isnan(run(True)).all()
    assert torch.equal(run(False), g0)          # the next call's forward zeroes its own word
