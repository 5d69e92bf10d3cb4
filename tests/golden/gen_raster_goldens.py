"""Generate ``raster_cases.npz``: small seeded rasterizer cases with the outputs of oracle O2
(fp32 tiled C restatement), cross-checked against oracle O1 (fp64 dense autograd) at generation time.

    python tests/golden/gen_raster_goldens.py

PARITY UNPINNED: these vectors come from THIS repo's oracles, not from the reference's CUDA extension
(absent, see oracle/gsr_oracle.c header).  They freeze the oracle so that later edits cannot silently
move the target, and give the GPU tests fixed inputs/outputs.  Sizes are deliberately not multiples of 16.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from util import random_gaussians, ring_camera, rel_err  # noqa: E402
from oracle import TiledOracle  # noqa: E402
from oracle.dense_oracle import dense_rasterize  # noqa: E402

CASES = [  # name, P, W, H, seed, bg, scale range
    ("p1_48x32", 1, 48, 32, 3, (0.2, 0.1, 0.3), (0.2, 0.4)),
    ("p8_48x32", 8, 48, 32, 4, (0, 0, 0), (0.1, 0.4)),
    ("p64_64x64", 64, 64, 64, 5, (0.5, 0.5, 0.5), (0.05, 0.4)),
    ("p512_70x100", 512, 70, 100, 6, (0, 0, 0), (0.02, 0.3)),
    ("p300_dense_100x70", 300, 100, 70, 7, (0.1, 0.2, 0.3), (0.1, 0.6)),  # heavy overlap: early termination
]


def main():
    out = {"names": np.array([c[0] for c in CASES])}
    for name, P, W, H, seed, bg, (slo, shi) in CASES:
        g = random_gaussians(P, seed=seed, scale_lo=slo, scale_hi=shi)
        cam = ring_camera(W, H, v=seed % 4, bg=bg)
        o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                         rotations=g["rotations"])
        dL = np.random.default_rng(100 + seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
        gr = o2.backward(dL)
        # cross-check against O1
        t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in g.items()}
        color, radii, depth, m2 = dense_rasterize(
            H, W, cam.tanfovx, cam.tanfovy, torch.tensor(cam.bg), 1.0, torch.tensor(cam.viewmatrix),
            torch.tensor(cam.projmatrix), 0, torch.tensor(cam.campos), t["means3D"], t["opacities"],
            colors_precomp=t["colors_precomp"], scales=t["scales"], rotations=t["rotations"])
        (color * torch.tensor(dL, dtype=torch.float64)).sum().backward()
        amb = o2.ambiguous
        e_col = np.abs(color.detach().numpy() - o2.color)[:, ~amb].max()
        errs = dict(color=e_col, radii=int((radii.numpy() != o2.radii).sum()),
                    means3D=rel_err(gr["means3D"], t["means3D"].grad.numpy()),
                    scales=rel_err(gr["scales"], t["scales"].grad.numpy()),
                    rot=rel_err(gr["rotations"], t["rotations"].grad.numpy()),
                    op=rel_err(gr["opacities"], t["opacities"].grad.numpy()),
                    col=rel_err(gr["colors_precomp"], t["colors_precomp"].grad.numpy()))
        print(name, "D=%d amb=%d" % (o2.num_rendered, amb.sum()), {k: float("%.2e" % v) for k, v in errs.items()})
        assert errs["radii"] == 0 and errs["color"] < 2e-5 and max(errs[k] for k in ("means3D", "scales", "rot", "op", "col")) < 2e-4
        pre = name + "/"
        for k, v in g.items():
            out[pre + "in_" + k] = v
        out[pre + "cam"] = np.concatenate([[H, W, cam.tanfovx, cam.tanfovy], cam.bg, cam.viewmatrix.reshape(-1),
                                           cam.projmatrix.reshape(-1), cam.campos]).astype(np.float64)
        out[pre + "dL_dcolor"] = dL
        out[pre + "color"] = o2.color
        out[pre + "depth"] = o2.depth
        out[pre + "radii"] = o2.radii
        out[pre + "point_list"] = o2.point_list
        out[pre + "ranges"] = o2.ranges
        out[pre + "n_contrib"] = o2.n_contrib
        out[pre + "final_T"] = o2.final_T
        out[pre + "ambiguous"] = amb
        out[pre + "means2D"] = o2.means2D
        out[pre + "conic_opacity"] = o2.conic_opacity
        for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
            out[pre + "grad_" + k] = gr[k]
    path = os.path.join(HERE, "raster_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
