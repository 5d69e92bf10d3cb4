"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/gsr.h
declares, the Python API has the reference's surface, and there is no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diff_gaussian_rasterization import _hip
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    declared = set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"gsr_settings", "gsr_debug_views", "gsr_kernel_time"}
    assert {"gsr_forward_preprocess", "gsr_forward_render", "gsr_backward", "gsr_mark_visible"} <= declared
    assert os.path.exists(_hip.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/gsr.h but not exported"
    assert set(_hip.EXPORTS) == declared


def test_host_only_entry_points_work_without_a_gpu():
    from diff_gaussian_rasterization import _hip
    lib = _hip.load_library()
    assert lib.gsr_version() == 124
    g1, g2 = lib.gsr_geom_bytes(1000), lib.gsr_geom_bytes(100000)
    assert 0 < g1 < g2 and g2 % 256 == 0
    assert lib.gsr_image_bytes(800, 800) >= 800 * 800 * 8 + 2500 * 8
    assert lib.gsr_binning_bytes(0, 800, 800) > 0
    assert lib.gsr_binning_bytes(460000, 800, 800) >= 460000 * (4 + 4 + 8 + 8 + 4)
    assert lib.gsr_backward_scratch_bytes(100000, 460000) >= 460000 * 36


def test_host_only_size_functions_of_the_step_kernels():
    """Sizes the caller allocates for the fused step pieces (no GPU needed): image-term block partials, shared-term work buffers."""
    from diff_gaussian_rasterization import _hip
    lib = _hip.load_library()
    per = lib.gsr_image_loss_blocks(1, 800, 800)
    assert per == 25 * 15                                          # 32 x 54 output tiles
    assert lib.gsr_image_loss_blocks(7, 33, 55) == 7 * 2 * 1
    assert lib.gsr_views_loss_blocks(8, 3, 800, 800) == 24 * per
    nfg, K, nbg = 70_000, 20, 30_000
    assert lib.gsr_shared_terms_scratch(nfg, K) == 16 * nfg + 8 * (nfg + nfg * K)
    assert lib.gsr_shared_terms_partials(nfg, nbg) >= 16 * nfg + 3 * lib.gsr_rigidity_blocks(nfg)
    assert lib.gsr_shared_terms_scratch(0, K) == 0 and lib.gsr_rigidity_blocks(0) == 0
    tab = _hip.GsrLossViews()
    assert len(tab.cam_row) == _hip.LOSS_MAX_IMAGES == 32


def test_settings_namedtuple_matches_reference_fields():
    """Field names/order constructed at /root/reference/src/tracking/helpers.py:20-32."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings as S
    assert S._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                         "projmatrix", "sh_degree", "campos", "prefiltered")
    s = S(image_height=4, image_width=5, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3), scale_modifier=1.0,
          viewmatrix=torch.eye(4)[None], projmatrix=torch.eye(4)[None], sh_degree=0, campos=torch.zeros(3),
          prefiltered=False)
    assert s.image_width == 5 and s._replace(sh_degree=2).sh_degree == 2


def _rast():
    from diff_gaussian_rasterization import GaussianRasterizationSettings as S, GaussianRasterizer
    s = S(8, 8, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4)[None], torch.eye(4)[None], 0, torch.zeros(3), False)
    return GaussianRasterizer(raster_settings=s)


def test_argument_validation_messages():
    r = _rast()
    z = torch.zeros
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), colors_precomp=z(2, 3), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), colors_precomp=z(2, 3), scales=z(2, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), colors_precomp=z(2, 3), scales=z(2, 3), rotations=z(2, 4), cov3D_precomp=z(2, 6))
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=z(2, 4), means2D=z(2, 3), opacities=z(2, 1), colors_precomp=z(2, 3), scales=z(2, 3), rotations=z(2, 4))


def test_no_cpu_fallback():
    """A CPU tensor must fail loudly, never render on the host."""
    r = _rast()
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), colors_precomp=z(2, 3), scales=z(2, 3), rotations=z(2, 4))


def test_missing_extension_fails_loudly(monkeypatch):
    from diff_gaussian_rasterization import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libgsr_hip.so")
    with pytest.raises(RuntimeError, match="HIP extension not found"):
        _hip.load_library()


def test_product_code_never_references_the_oracle():
    pkg = os.path.join(ROOT, "gs-dynamics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libgsr_oracle" not in txt, f


def test_autograd_glue_with_test_double(monkeypatch):
    """Reference call pattern on CPU through the real autograd glue (backend replaced by the test double):
    means2D is a non-leaf 'zeros + 0' holder with retain_grad, as /root/reference/src/tracking/helpers.py:43."""
    import oracle_double
    oracle_double.install(monkeypatch)
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(60, device="cpu", scale_lo=0.05, scale_hi=0.3)
    cam = synth_ring_cameras(4, 48, 32, device="cpu")[0]
    rv = params2rendervar(params)
    rv["means2D"].retain_grad()
    im, radius, depth = GaussianRasterizer(raster_settings=cam)(**rv)
    assert im.shape == (3, 32, 48) and depth.shape == (1, 32, 48) and radius.dtype == torch.int32
    (im.sum() + 0.0 * depth.sum()).backward()   # a gradient on depth is accepted and ignored
    assert rv["means2D"].grad is not None and rv["means2D"].grad.shape == (60, 3)
    assert params["means3D"].grad is not None and params["log_scales"].grad is not None
    assert params["rgb_colors"].grad is None     # requires_grad False in the reference (train_utils.py:133)


def test_wait_counts_is_a_host_only_poll():
    """gsr_wait_counts (ABI 117): spins on a host array from C -- returns the maximum once every word is >= 0 (a second thread plays the
    tile-order kernel's system-scope stores), -1 on timeout; no device involved."""
    import ctypes
    import threading
    import time
    import numpy as np
    from diff_gaussian_rasterization import _hip
    lib = _hip.load_library()
    a = np.full(4, -1, np.int32)
    assert lib.gsr_wait_counts(a.ctypes.data, 4, 50, 2000) == -1              # nobody writes: timeout after ~2 ms
    def writer():
        time.sleep(0.02)
        a[:3] = (7, 123456, 5)
        time.sleep(0.01)
        a[3] = 9
    t = threading.Thread(target=writer)
    t.start()
    assert lib.gsr_wait_counts(a.ctypes.data, 4, 50, 2_000_000) == 123456       # ctypes released the GIL: the writer thread ran
    t.join()
    assert lib.gsr_wait_counts(None, 0, 0, 0) == 0
