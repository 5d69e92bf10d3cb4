import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gs-dynamics_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- fixtures of the GPU parity tests (-m gpu)
@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def full_scene(dev):
    """BASELINE.json's scene: 100 k Gaussians, four 800x800 ring cameras."""
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H = 100_000, 800, 800
    params = synth_scene_params(P, device=dev)
    cams = synth_ring_cameras(4, W, H, device=dev)
    return params, cams, params2rendervar
