"""Generate ``densify_host.npz``: golden vectors of the reference's adaptive density control
(/root/reference/src/tracking/external.py:138-299), captured by importing it here on the CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_densify_goldens.py

Inputs (seeded parameters, Adam moments after one step, gradient accumulators) and outputs (parameters, moments,
accumulators after the call) at iterations 600 (clone + split + prune) and 3000 (prune large + opacity reset).
Only data is stored."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "densify_host.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("open3d", "cv2", "dgl", "ipdb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    torch.Tensor.cuda = lambda self, *a, **k: self
    o_zeros, o_tensor, o_zeros_like = torch.zeros, torch.tensor, torch.zeros_like

    def strip(fn):
        def inner(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    torch.zeros, torch.tensor, torch.zeros_like = strip(o_zeros), strip(o_tensor), strip(o_zeros_like)
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, os.path.join(REF, "tracking"))
    import external  # noqa
    return external


def make_problem(P=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    params = {
        "means3D": torch.randn(P, 3, generator=g),
        "rgb_colors": torch.rand(P, 3, generator=g),
        "seg_colors": torch.rand(P, 3, generator=g),
        "unnorm_rotations": torch.randn(P, 4, generator=g),
        "logit_opacities": torch.randn(P, 1, generator=g) * 3.0,
        "log_scales": torch.log(torch.rand(P, 3, generator=g) * 0.3 + 0.005),
        "cam_m": torch.zeros(5, 3), "cam_c": torch.zeros(5, 3),
    }
    params = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    lrs = {"means3D": 1e-3, "rgb_colors": 0.0, "seg_colors": 0.0, "unnorm_rotations": 1e-3, "logit_opacities": 0.05,
           "log_scales": 1e-3, "cam_m": 1e-4, "cam_c": 1e-4}
    opt = torch.optim.Adam([{"params": [v], "name": k, "lr": lrs[k]} for k, v in params.items()], lr=0.0, eps=1e-15)
    for k, v in params.items():                      # one step so that every group has Adam moments
        v.grad = torch.randn(v.shape, generator=g) * 0.01
    opt.step()
    opt.zero_grad(set_to_none=True)
    variables = {
        "means2D_gradient_accum": torch.rand(P, generator=g) * 0.002,
        "denom": torch.randint(0, 6, (P,), generator=g).float(),
        "max_2D_radius": torch.rand(P, generator=g) * 20,
        "scene_radius": 2.0,
        "seen": torch.rand(P, generator=g) > 0.3,
    }
    m2 = torch.zeros(P, 3, requires_grad=True)
    m2.grad = torch.randn(P, 3, generator=g) * 1e-3
    variables["means2D"] = m2
    return params, variables, opt


def dump(out, tag, params, variables, opt):
    for k, v in params.items():
        out[f"{tag}_p_{k}"] = v.detach().numpy().copy()
        st = opt.state.get(v, None)
        if st is not None:
            out[f"{tag}_m_{k}"] = st["exp_avg"].numpy().copy()
            out[f"{tag}_v_{k}"] = st["exp_avg_sq"].numpy().copy()
    for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
        out[f"{tag}_{k}"] = variables[k].numpy().copy()


def main():
    ext = import_reference()
    out = {}
    for it in (600, 3000):
        params, variables, opt = make_problem(seed=it)
        out[f"i{it}_seen"] = variables["seen"].numpy().copy()
        out[f"i{it}_m2grad"] = variables["means2D"].grad.numpy().copy()
        dump(out, f"i{it}_in", params, variables, opt)
        torch.manual_seed(1234)
        with torch.no_grad():
            params, variables, n = ext.densify(params, variables, opt, it, 0.005, 0.25, 0.05)
        dump(out, f"i{it}_out", params, variables, opt)
        out[f"i{it}_n"] = np.array([n])
        print(it, "->", n, "points")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
