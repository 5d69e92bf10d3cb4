"""Debug helper (GPU box): replay ONE case of tests/test_soak_gpu.py and print what the list check compares.  python tests/soak_repro.py CASE [SEED]"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "gs-dynamics_amd")):
    sys.path.insert(0, p)
from hipcheck import _run_hip
from util import oracle_camera, random_gaussians, look_at
from oracle import TiledOracle
want, seed0 = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 77
rng = np.random.default_rng(seed0)
for case in range(want + 1):
    P = int(rng.choice([1, 5, 40, 150, 600, 1500, 4000]))
    W, H = int(rng.integers(8, 260)), int(rng.integers(8, 200))
    lo = float(rng.choice([0.003, 0.02, 0.08]))
    hi = lo * float(rng.choice([1.5, 8.0, 30.0]))
    kind = str(rng.choice(["rgb", "rgb", "sh", "cov3d"]))
    deg = int(rng.integers(0, 4))
    g = random_gaussians(P, seed=seed0 * 1000 + case, scale_lo=lo, scale_hi=hi, spread=float(rng.choice([0.4, 1.0, 2.0])), sh_M=16 if kind == "sh" else 0)
    shift = float(rng.choice([-2.5, 0.0, 2.0]))
    g["opacities"] = (1.0 / (1.0 + np.exp(-(np.log(g["opacities"] / (1.0 - g["opacities"])) + shift)))).astype(np.float32)
    ang, rad, hgt = float(rng.uniform(0, 6.28)), float(rng.choice([0.7, 2.0, 4.0, 8.0])), float(rng.choice([-0.6, 0.5, 2.5]))
    f = float(rng.choice([0.6, 1.0, 1.8])) * W
    cam = oracle_camera(W, H, look_at((rad * np.cos(ang), hgt, rad * np.sin(ang))), fx=f, fy=f * float(rng.choice([1.0, 1.2])),
                        cx=W / 2 + float(rng.choice([0.0, 0.0, 0.13 * W])), cy=H / 2 - float(rng.choice([0.0, 0.09 * H])),
                        bg=tuple(float(x) for x in rng.uniform(0, 1, 3)), sh_degree=deg if kind == "sh" else 0)
print("case", want, kind, "P", P, W, H, "scales", lo, hi)
dev = torch.device("cuda:0")
o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g.get("colors_precomp"), scales=g.get("scales"), rotations=g.get("rotations"), nthreads=4)
color, radii, depth, grads, views = _run_hip(cam, g, dev, dL=None, want_state=True)
np.set_printoptions(precision=6, suppress=True, linewidth=200)
print("oracle radii", o2.radii, "hip radii", radii)
print("oracle means2D", o2.means2D); print("oracle conic_opacity", o2.conic_opacity)
print("oracle tiles_touched", o2.tiles_touched, "num_rendered", o2.num_rendered)
print("hip tiles_touched", views["tiles_touched"].cpu().numpy(), "offsets", views["offsets"].cpu().numpy())
rect = views["rect"].cpu().numpy().astype(np.uint32)
print("hip rect x0,y0,x1,y1", np.stack([rect[:, 0] & 0xffff, rect[:, 0] >> 16, rect[:, 1] & 0xffff, rect[:, 1] >> 16], 1))
print("oracle rect", o2.rect)
print("hip rec word 15 (mask)", [hex(int(x) & 0xffffffff) for x in views["rec"][:, 15].contiguous().view(torch.int32).cpu().numpy()])
print("hip rec", views["rec"].cpu().numpy())
print("oracle ranges nonzero", [(t, tuple(r)) for t, r in enumerate(o2.ranges) if r[1] > r[0]], "point_list", o2.point_list)
print("hip ranges nonzero", [(t, tuple(r)) for t, r in enumerate(views["ranges"].cpu().numpy()) if r[1] > r[0]], "point_list", views["point_list"].cpu().numpy())
