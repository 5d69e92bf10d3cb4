"""Generate ``raster_cases_views.npz``: multi-view and shared-camera ("pair") cases of the rasterizer for the batched
entry point (``rasterize_gaussians_views``), with the outputs of oracle O2 run view by view.

    python tests/golden/gen_raster_view_goldens.py

PARITY UNPINNED, like raster_cases.npz: the vectors come from THIS repo's oracle (the reference's CUDA extension is
absent).  What they pin: per-view images / radii / depth, per-view screen-space gradients, per-view colour gradients where
views carry their own colours, and the SUM over views of every other input gradient -- the contract of the multi-view call
(the reference renders one view per call and lets autograd add the gradients up, train_utils.py:174-195).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from util import random_gaussians, ring_camera  # noqa: E402
from oracle import TiledOracle  # noqa: E402

# name, P, W, H, seed, scale range, views as (camera index on the ring, colour set), number of colour sets
CASES = [
    ("mv3_p400_90x66", 400, 90, 66, 11, (0.03, 0.3), [(0, 0), (1, 0), (3, 0)], 1),
    ("pairs_p300_70x52", 300, 70, 52, 12, (0.04, 0.35), [(2, 0), (2, 1), (0, 0), (0, 1)], 2),   # colour + seg render per camera
]
SUMMED = ("means3D", "opacities", "scales", "rotations")


def cam_vec(cam):
    return np.concatenate([[cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy], cam.bg, cam.viewmatrix.reshape(-1),
                           cam.projmatrix.reshape(-1), cam.campos]).astype(np.float64)


def main():
    out = {"names": np.array([c[0] for c in CASES])}
    for name, P, W, H, seed, (slo, shi), views, ncol in CASES:
        g = random_gaussians(P, seed=seed, scale_lo=slo, scale_hi=shi)
        rng = np.random.default_rng(500 + seed)
        colours = [g["colors_precomp"]] + [rng.uniform(0, 1, (P, 3)).astype(np.float32) for _ in range(ncol - 1)]
        pre = name + "/"
        for k in ("means3D", "scales", "rotations", "opacities"):
            out[pre + "in_" + k] = g[k]
        out[pre + "in_colours"] = np.stack(colours)                         # [ncol, P, 3]
        out[pre + "view_cam"] = np.array([v[0] for v in views], np.int32)   # ring index per view (equal index = same camera)
        out[pre + "view_colour"] = np.array([v[1] for v in views], np.int32)
        sums = {k: np.zeros_like(g[k], dtype=np.float64) for k in SUMMED}
        col_grads = np.zeros((len(views), P, 3), np.float64)
        per = {k: [] for k in ("cam", "dL_dcolor", "color", "depth", "radii", "ambiguous", "grad_means2D")}
        for vi, (ci, col) in enumerate(views):
            cam = ring_camera(W, H, v=ci, bg=(0.1 * ci, 0.05, 0.2))
            o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=colours[col], scales=g["scales"],
                             rotations=g["rotations"])
            dL = rng.uniform(-1, 1, (3, H, W)).astype(np.float32)
            dL[:, o2.ambiguous] = 0.0
            gr = o2.backward(dL)
            for k in SUMMED:
                sums[k] += gr[k].astype(np.float64)
            col_grads[vi] = gr["colors_precomp"]
            for k, v in (("cam", cam_vec(cam)), ("dL_dcolor", dL), ("color", o2.color), ("depth", o2.depth), ("radii", o2.radii),
                         ("ambiguous", o2.ambiguous), ("grad_means2D", gr["means2D"])):
                per[k].append(v)
            print(name, "view", vi, "D=%d ambiguous=%d" % (o2.num_rendered, int(o2.ambiguous.sum())))
        for k, v in per.items():
            out[pre + k] = np.stack(v)
        for k in SUMMED:
            out[pre + "grad_sum_" + k] = sums[k].astype(np.float32)
        out[pre + "grad_colours_per_view"] = col_grads.astype(np.float32)
    path = os.path.join(HERE, "raster_cases_views.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
