"""Debug helper (GPU box): the worst non-ambiguous pixel of one soak case (colour / depth / final_T against the fp32 oracle) and every
threshold decision of the oracle's walk of that pixel that sits near its threshold.  python tests/soak_pixel_diag.py CASE SEED"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "soak_diag.py")).read()
exec(src[:src.index("g32, g64 = o32.backward(dL)")])
color, radii, depth, _, _ = _run_hip(cam, g, dev, dL=None)
err = np.abs(depth[0] - o32.depth[0]) / (np.abs(o32.depth[0]) + 1e-4 * np.abs(o32.depth).max())
err[~ok] = 0
py, px = np.unravel_index(int(err.argmax()), err.shape)
print("worst depth pixel", (px, py), "hip", depth[0, py, px], "oracle", o32.depth[0, py, px], "rel", err[py, px], "| colour hip", color[:, py, px], "oracle", o32.color[:, py, px],
      "| n_contrib", int(o32.n_contrib[py, px]), "final_T", float(o32.final_T[py, px]))
f = np.float32
tile = (py // 16) * ((W + 15) // 16) + px // 16
r0, r1 = o32.ranges[tile]
T = f(1.0)
for s in range(int(r0), int(r1)):
    gi = int(o32.point_list[s])
    dx, dy = f(o32.means2D[gi, 0]) - f(px), f(o32.means2D[gi, 1]) - f(py)
    A, B, C, op = (f(v) for v in o32.conic_opacity[gi])
    power = f(f(-0.5) * f(f(f(A * dx) * dx) + f(f(C * dy) * dy))) - f(f(B * dx) * dy)
    S = 0.5 * abs(A) * dx * dx + 0.5 * abs(C) * dy * dy + abs(B * dx * dy)
    p64 = -0.5 * (float(A) * float(dx) ** 2 + float(C) * float(dy) ** 2) - float(B) * float(dx) * float(dy)
    if power > 0:
        if abs(power) < 1e-3: print(f"  entry {s - r0} g {gi}: power {power:.3e} > 0 (fp64 {p64:.3e}) terms {S:.1f} opacity {op:.3f}")
        continue
    alpha = min(f(0.99), f(op * np.exp(power)))
    if abs(alpha * 255 - 1) < 2e-3 or abs(power) < 1e-3:
        print(f"  entry {s - r0} g {gi}: alpha*255-1 = {alpha * 255 - 1:.3e} (fp64 {op * np.exp(p64) * 255 - 1:.3e}) power {power:.4f} terms {S:.1f} slack {4 * 5.96e-8 * S:.2e} T {T:.3e}")
    if alpha < f(1 / 255): continue
    tT = f(T * f(1 - alpha))
    if abs(tT / 1e-4 - 1) < 1e-2: print(f"  entry {s - r0} g {gi}: test_T {tT:.6e} near 1e-4, alpha {alpha:.4f} depth {o32.depths[gi] if hasattr(o32, 'depths') else '?'}")
    if tT < f(1e-4): break
    T = tT
