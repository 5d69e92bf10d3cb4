"""Debug helper (GPU box): for one case of the soak and one Gaussian, which TILE's pixels carry the difference between the HIP backward and
the oracle?  dL is restricted to one tile at a time.  python tests/soak_tile_diag.py CASE SEED ROW"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "soak_diag.py")).read()
row = int(sys.argv[3]); sys.argv = sys.argv[:3]
exec(src[:src.index("g32, g64 = o32.backward(dL)")])     # the case, the oracles, dL (soak_diag.py's own construction)
gx, gy = (W + 15) // 16, (H + 15) // 16
print("tiles", gx, "x", gy, "row", row)
bad = []
for ty in range(gy):
    for tx in range(gx):
        d = np.zeros_like(dL)
        d[:, 16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16] = dL[:, 16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16]
        go = o32.backward(d)
        _, _, _, gh, _ = _run_hip(cam, g, dev, dL=d)
        a, b = gh["colors_precomp"][row], go["colors_precomp"][row]
        e = float(np.abs(a - b).max())
        flag = e > 1e-5 * max(1e-3, float(np.abs(b).max()))
        print(f"tile ({tx},{ty}): colour grad hip {a} oracle {b} {'<-- DIFFERS' if flag else ''}")
        if flag:
            bad.append((tx, ty))
for tx, ty in bad[:2]:
    print("pixels of tile", tx, ty)
    for py in range(16 * ty, min(H, 16 * ty + 16)):
        for px in range(16 * tx, min(W, 16 * tx + 16)):
            d = np.zeros_like(dL); d[:, py, px] = dL[:, py, px]
            if not np.any(d):
                continue
            go = o32.backward(d)
            _, _, _, gh, _ = _run_hip(cam, g, dev, dL=d)
            a, b = gh["colors_precomp"][row], go["colors_precomp"][row]
            if float(np.abs(a - b).max()) > 1e-5 * max(1e-6, float(np.abs(b).max())):
                print(f"  pixel ({px},{py}): hip {a} oracle {b}  n_contrib {int(o32.n_contrib[py, px])} final_T {float(o32.final_T[py, px]):.3e}")
