"""Debug helper (GPU box): replay ONE case of tests/test_soak_gpu.py and print what the list check compares.  python tests/soak_repro.py CASE [SEED]"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "gs-dynamics_amd")):
    sys.path.insert(0, p)
from hipcheck import _run_hip
from oracle import TiledOracle
from soak_cases import case_at
want, seed0 = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 77
cam, g, tag, c = case_at(seed0, want)
P, W, H, lo, hi, kind = c["P"], c["W"], c["H"], c["lo"], c["hi"], c["kind"]
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
