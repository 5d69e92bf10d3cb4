"""Debug helper (GPU box): one case of tests/test_soak_gpu.py against the fp32 AND the fp64 oracle, every gradient tensor: norm-wise and
worst-row distance of the HIP path and of the fp32 oracle from fp64.  python tests/soak_diag.py CASE [SEED]"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "gs-dynamics_amd")):
    sys.path.insert(0, p)
from hipcheck import _run_hip
from util import rel_err, row_err
from oracle import TiledOracle
from soak_cases import case_at
want, seed0 = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 77
cam, g, tag, c = case_at(seed0, want, os.environ.get("GSR_SOAK_BIG") == "1")
P, W, H, lo, hi, kind = c["P"], c["W"], c["H"], c["lo"], c["hi"], c["kind"]
print("case", want, kind, "P", P, W, H, "scales", lo, hi)
dev = torch.device("cuda:0")
kw = dict(colors_precomp=g.get("colors_precomp"), shs=g.get("shs"), scales=g.get("scales"), rotations=g.get("rotations"), cov3D_precomp=g.get("cov3D_precomp"), nthreads=4)
o32 = TiledOracle(cam, g["means3D"], g["opacities"], **kw)
o64 = TiledOracle(cam, g["means3D"], g["opacities"], f64=True, decisions_of=o32, **kw)
ok = ~o32.ambiguous
dL = np.random.default_rng(want).uniform(-1, 1, (3, H, W)).astype(np.float32)
dL[:, ~ok] = 0.0
g32, g64 = o32.backward(dL), o64.backward(dL)
_, _, _, grads, _ = _run_hip(cam, g, dev, dL=dL)
np.set_printoptions(precision=4, linewidth=220)
print("radii", o32.radii.tolist(), "means2D range", o32.means2D.min(0), o32.means2D.max(0))
for k, v in grads.items():
    e_h, e_o = rel_err(v, g64[k]), rel_err(g32[k], g64[k])
    (r_h, i_h), (r_o, i_o) = row_err(v, g64[k]), row_err(g32[k], g64[k])
    print(f"{k:16s} norm-wise HIP {e_h:.2e} oracle32 {e_o:.2e} | worst row HIP {r_h:.2e} (row {i_h}) oracle32 {r_o:.2e} (row {i_o}) | max|g| {np.abs(g64[k]).max():.3e}")
k = "means3D"
i = int(np.abs(grads[k] - g64[k]).max(1).argmax())
print("worst means3D row", i, "hip", grads[k][i], "o32", g32[k][i], "o64", g64[k][i], "| radius", int(o32.radii[i]), "mean2D", o32.means2D[i], "conic_op", o32.conic_opacity[i])
print("   means2D grad of that row: hip", grads["means2D"][i], "o32", g32["means2D"][i], "o64", g64["means2D"][i])
if "cov3D_precomp" in grads:
    print("   cov3D grad of that row: hip", grads["cov3D_precomp"][i], "\n      o32", g32["cov3D_precomp"][i], "\n      o64", g64["cov3D_precomp"][i])

for k2 in grads:
    if k2 not in ("means3D", "means2D", "cov3D_precomp") and grads[k2].ndim >= 2:
        print(f"   {k2} of that row: hip", grads[k2][i].reshape(-1)[:9], "o32", g32[k2][i].reshape(-1)[:9], "o64", g64[k2][i].reshape(-1)[:9])
print("   means2D grad of that row: hip", grads["means2D"][i], "o32", g32["means2D"][i], "o64", g64["means2D"][i])
print("   tiles_touched", o32.tiles_touched[i], "rect", o32.rect[i], "depth", o32.depths[i] if hasattr(o32, "depths") else None)
