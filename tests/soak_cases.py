"""The parity soak's seeded case stream (tests/test_soak_gpu.py), shared with its regression cases and the soak_*_diag.py helpers.
``draw_case`` consumes the stream's random numbers of ONE case (cheap: no scene is built), ``build_case`` builds the scene of a drawn
case, ``case_at`` = the two for case number ``want`` of stream ``seed0``."""
import numpy as np

from oracle import TiledOracle
from util import look_at, oracle_camera, random_gaussians


def draw_case(rng, big=False):
    P = int(rng.choice([1, 5, 40, 150, 600, 1500, 4000]))
    W, H = int(rng.integers(8, 260)), int(rng.integers(8, 200))
    if big:        # GSR_SOAK_BIG=1: the same stream of choices at 10 - 40 x the Gaussians and ~3 x the image side (long lists, every sort build)
        P, W, H = int(rng.choice([8000, 20000, 50000])), 3 * W + 5, 3 * H + 3
    lo = float(rng.choice([0.003, 0.02, 0.08]))
    hi = lo * float(rng.choice([1.5, 8.0, 30.0]))
    kind = str(rng.choice(["rgb", "rgb", "sh", "cov3d"]))
    deg = int(rng.integers(0, 4))
    spread = float(rng.choice([0.4, 1.0, 2.0]))
    shift = float(rng.choice([-2.5, 0.0, 2.0]))
    ang, rad, hgt = float(rng.uniform(0, 6.28)), float(rng.choice([0.7, 2.0, 4.0, 8.0])), float(rng.choice([-0.6, 0.5, 2.5]))
    f = float(rng.choice([0.6, 1.0, 1.8])) * W
    fy = f * float(rng.choice([1.0, 1.2]))
    cx = W / 2 + float(rng.choice([0.0, 0.0, 0.13 * W]))
    cy = H / 2 - float(rng.choice([0.0, 0.09 * H]))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    return dict(P=P, W=W, H=H, lo=lo, hi=hi, kind=kind, deg=deg, spread=spread, shift=shift, ang=ang, rad=rad, hgt=hgt, f=f, fy=fy, cx=cx, cy=cy, bg=bg)


def build_case(seed0, case, c):
    """-> (cam, g, tag): the oracle camera, the Gaussians' input dict, the case's one-line description."""
    g = random_gaussians(c["P"], seed=seed0 * 1000 + case, scale_lo=c["lo"], scale_hi=c["hi"], spread=c["spread"], sh_M=16 if c["kind"] == "sh" else 0)
    g["opacities"] = (1.0 / (1.0 + np.exp(-(np.log(g["opacities"] / (1.0 - g["opacities"])) + c["shift"])))).astype(np.float32)
    cam = oracle_camera(c["W"], c["H"], look_at((c["rad"] * np.cos(c["ang"]), c["hgt"], c["rad"] * np.sin(c["ang"]))), fx=c["f"], fy=c["fy"],
                        cx=c["cx"], cy=c["cy"], bg=c["bg"], sh_degree=c["deg"] if c["kind"] == "sh" else 0)
    if c["kind"] == "sh":
        del g["colors_precomp"]
    elif c["kind"] == "cov3d":
        probe = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"], rotations=g["rotations"])
        g = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=g["colors_precomp"], cov3D_precomp=probe.cov3D)
    tag = (f"case {case}: {c['kind']}{c['deg'] if c['kind'] == 'sh' else ''} P={c['P']} {c['W']}x{c['H']} scales {c['lo']}..{c['hi']:.3f} "
           f"cam r={c['rad']} h={c['hgt']}")
    return cam, g, tag


def case_at(seed0, want, big=False):
    rng = np.random.default_rng(seed0)
    for _ in range(want):
        draw_case(rng, big)
    c = draw_case(rng, big)
    return (*build_case(seed0, want, c), c)
