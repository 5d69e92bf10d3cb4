"""Parity soak (round 5): a long seeded stream of random (scene, camera, image size, colour model) combinations through the drop-in module
against the CPU oracle, with the full check of ``hipcheck._check_against_oracle`` -- radii and tile lists bit-exact, images 1e-4, every
gradient norm-wise 1e-4 and row-wise.  A different stream from ``test_randomised_sweep_vs_oracle`` (other seeds, SH degrees 0 - 3 and
precomputed 3D covariances mixed in, off-centre principal points).  ``GSR_SOAK_CASES`` sets the length (default 200: ~4 s on the GPU box; the round's
long runs used 600 - 1500 per stream -- profiles/r05_parity_soak.txt) and ``GSR_SOAK_SEED`` the stream."""
import os

import numpy as np
import pytest

from hipcheck import ROW_TOL_WORST_P5000, TOL, _check_against_oracle, _run_hip
from soak_cases import build_case, case_at, draw_case
from util import mixed_err, rel_err, row_err
from oracle import TiledOracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _adjudicate(cam, g, dev, seed, tol_worst):
    """A gradient off by more than the bar against the fp32 oracle: whose error is it?  The fp64 build of the oracle (taking over the fp32
    run's discrete decisions) is the referee -- tiny scenes (one or two Gaussians) have gradients that are small differences of large
    per-pixel terms, and so have single rows of larger scenes (Gaussians bigger than the scene above all): BOTH fp32 evaluations then sit
    1e-4 .. 1e-3 from the exact value.  Accepted as conditioning when the HIP path is no further from fp64 than twice the fp32 oracle is
    (+ 2e-5) norm-wise (test_row_wise_error_against_the_fp64_oracle's rule) and four times (+ 1e-4) in its worst row -- ONE row of
    thousands, measured relative to that row's own gradient (1600 cases: the HIP path's worst row is 0.5 ... 3.9 x the fp32 oracle's)."""
    kw = dict(colors_precomp=g.get("colors_precomp"), shs=g.get("shs"), scales=g.get("scales"), rotations=g.get("rotations"),
              cov3D_precomp=g.get("cov3D_precomp"), nthreads=4)
    o32 = TiledOracle(cam, g["means3D"], g["opacities"], **kw)
    o64 = TiledOracle(cam, g["means3D"], g["opacities"], f64=True, decisions_of=o32, **kw)
    ok = ~o32.ambiguous
    dL = np.random.default_rng(seed).uniform(-1, 1, (3, cam.image_height, cam.image_width)).astype(np.float32)
    dL[:, ~ok] = 0.0
    g32, g64 = o32.backward(dL), o64.backward(dL)
    color, _, depth, grads, views = _run_hip(cam, g, dev, dL=dL, want_state=True)
    notes, rows = [], []
    over = float(o32.radii.max()) / float(np.hypot(cam.image_width, cam.image_height))     # > 1: some Gaussian's 3-sigma radius exceeds the image diagonal
    # final transmittance: a product of up to hundreds of (1 - alpha) factors, each carrying alpha's absolute rounding error -- 1e-5 RELATIVE where
    # alpha sits at the 0.99 clamp -- so where T is small two fp32 evaluations differ by more than 1e-4 of it (the images do not: T only ever
    # enters them absolutely).  Refereed like the gradients; pixels where the builds DECIDED differently are not compared (as everywhere).
    okT = ok & ~o64.ambiguous & (o32.n_contrib == o64.n_contrib)
    if okT.any():
        tT_h, tT_o = mixed_err(views["final_T"].cpu().numpy()[okT], o64.final_T[okT].astype(np.float32)), mixed_err(o32.final_T[okT], o64.final_T[okT].astype(np.float32))
        notes.append(f"final_T: vs fp64 HIP {tT_h:.2e} / fp32 oracle {tT_o:.2e}")
        if not tT_h <= max(TOL, 2.0 * tT_o + 2e-5):
            raise AssertionError(("final_T", tT_h, tT_o, over))      # (raised by hand: a plain tuple in args, no assertion rewriting)
    for name, img_h, img_o, img_64 in (("colour", color, o32.color, o64.color), ("depth", depth, o32.depth, o64.depth)):
        i_h, i_o = mixed_err(img_h[:, okT], img_64[:, okT].astype(np.float32)), mixed_err(img_o[:, okT], img_64[:, okT].astype(np.float32))
        notes.append(f"{name}: vs fp64 HIP {i_h:.2e} / fp32 oracle {i_o:.2e}")
        if not i_h <= max(TOL, 2.0 * i_o + 2e-5):
            raise AssertionError((name + " vs fp64", i_h, i_o, over))
    for k, v in grads.items():
        e_hip, e_o = rel_err(v, g64[k]), rel_err(g32[k], g64[k])
        r_hip, r_o = row_err(v, g64[k])[0], row_err(g32[k], g64[k])[0]
        notes.append(f"{k}: vs fp64 norm-wise HIP {e_hip:.2e} / fp32 oracle {e_o:.2e}, worst row HIP {r_hip:.2e} / fp32 oracle {r_o:.2e}")
        if not e_hip <= max(TOL, 2.0 * e_o + 2e-5):
            raise AssertionError((k, e_hip, e_o, over))
        if not r_hip <= max(tol_worst, 4.0 * r_o + 1e-4):      # ONE row of the tensor, relative to its own gradient: tallied, not fatal (see the caller)
            rows.append(f"{k}: worst row HIP {r_hip:.2e} / fp32 oracle {r_o:.2e}")
    return "; ".join(notes), rows


def test_parity_soak(dev):
    n_cases, seed0 = int(os.environ.get("GSR_SOAK_CASES", "200")), int(os.environ.get("GSR_SOAK_SEED", "77"))
    rng = np.random.default_rng(seed0)
    big = os.environ.get("GSR_SOAK_BIG") == "1"
    done, skipped, conditioned, kinds = 0, 0, 0, {"rgb": 0, "sh": 0, "cov3d": 0}
    missed, row_missed, seen_ref = [], [], 0
    log = os.path.join(os.path.dirname(HERE), "gpurun_out", "parity_soak.txt")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "a") as fh:
        fh.write(f"# parity soak: {n_cases} cases, seed {seed0}\n")
        for case in range(n_cases):
            c = draw_case(rng, big)
            cam, g, tag = build_case(seed0, case, c)
            hi, kind = c["hi"], c["kind"]
            tol_worst = 1e-3 if hi >= 1.0 else ROW_TOL_WORST_P5000
            try:
                try:
                    _check_against_oracle(cam, g, dev, seed=case, min_ok=0.98, tol_worst=tol_worst)
                except AssertionError as e:
                    # gradient bars and the transmittance go to the referee, and so does a colour / depth pixel off by more than 1e-4 OF ITSELF (a nearly
                    # empty pixel: the sum of dozens of alpha ~ 1/255 contributions of huge Gaussians, each alpha carrying the rounding of a cancelling
                    # exponent -- seed 4242 case 933: depth 0.0177 at final_T 0.992, 1.2e-4 apart); integers and ambiguous-pixel bounds never
                    if not (str(e).startswith(("grad ", "oracle P=", "final_T")) or str(e) in ("depth", "colour")):
                        raise
                    note, rows = _adjudicate(cam, g, dev, case, tol_worst)
                    conditioned += 1
                    fh.write(tag + f": fp32 bar missed ({e}) -- fp64 referee: {note}\n")
                    if rows:     # norm-wise within the rule, a single row beyond it: one Gaussian (typically centred off a very small image) whose
                        row_missed.append(tag)     # gradient is a small difference of large per-pixel terms -- counted, bounded below
                        fh.write(tag + ": WORST ROW beyond the referee's rule -- " + "; ".join(rows) + "\n")
                done += 1
                kinds[kind] += 1
            except AssertionError as e:
                if "too many threshold-ambiguous pixels" in str(e):     # a few huge faint Gaussians: nothing to compare tightly
                    skipped += 1
                    fh.write(tag + ": skipped (threshold-ambiguous scene)\n")
                    continue
                # A miss even by the referee's rule.  Scenes whose Gaussians are LARGER THAN THE SCENE (scale up to 2.4 in a unit cloud: alpha at
                # the 0.99 clamp over most of the image, quadratic forms that are differences of terms in the hundreds) are recorded and counted --
                # they are outside what the bars are stated for and the long runs exist to show how often they occur;
                # anywhere else a miss fails the test on the spot.
                # (GSR_SOAK_BIG=1, the exploratory size class: every miss is recorded and counted -- at three times the image side and 50 k Gaussians a
                # scale of 0.6 is a radius of 100+ pixels on a 60-pixel-wide image -- and the run fails on their NUMBER, below)
                info = e.args[0] if e.args and isinstance(e.args[0], tuple) else ()
                over = info[-1] if len(info) == 4 and isinstance(info[-1], float) else 0.0
                # ... and so is a scene in which some Gaussian's 3-sigma radius exceeds the IMAGE DIAGONAL (a camera inside the cloud, 0.25 in front of
                # a Gaussian of scale 0.6: radius 804 px on a 109 x 171 image, seed 123 case 1020 -- the same class seen from the image's side)
                if hi < 1.0 and not big and not over > 1.0:
                    fh.write(tag + f": FAILED {e}\n")
                    raise AssertionError(f"{tag}: {e}") from e
                missed.append(tag)
                fh.write(tag + (": BAR MISSED in a scene of Gaussians larger than the scene -- " if hi >= 1.0 else f": BAR MISSED with a Gaussian larger than the image (radius / diagonal {over:.1f}) -- " if over > 1.0 else ": BAR MISSED (big size class) -- ") + f"{str(e)[:300]}\n")
                continue
            if conditioned == seen_ref:
                fh.write(tag + ": ok\n")
            seen_ref = conditioned
        fh.write(f"# passed {done} (of which {conditioned} through the fp64 referee), skipped {skipped}, bars missed in {len(missed)} scenes of Gaussians larger "
                 f"than the scene, a single row beyond the referee's rule in {len(row_missed)}, of {n_cases}; by colour model {kinds}\n")
    assert done >= 0.8 * n_cases and conditioned <= (0.3 if big else 0.1) * n_cases + 2 and len(missed) <= 0.02 * n_cases + 1 and len(row_missed) <= 0.01 * n_cases + 1, (done, skipped, conditioned, missed, row_missed)


# Cases the long streams stopped at in round 5, as fixed regression cases of every -m gpu run.  Both were the blend kernels' conic rounded once
# per Gaussian after scaling by log2 e, on a nearly singular conic (profiles/r05_conic_prescale_precision.txt); since round 6 the conic is
# staged with its exact factors only (csrc/gsr_render.hip: gsr_power):
#   seed 77 case 671      sh0, 40 Gaussians, 127x123: `scales` 1.17e-4 norm-wise from fp64 (fp32 oracle 2.5e-5)   -> 1.9e-5
#   BIG seed 6 case 23    50 000 Gaussians of scale up to 0.6 on a 59x429 image: rotations 2.7e-4 (oracle 9.7e-5)   -> 9.7e-5
# The bar here is the soak's own, with nothing tallied: the fp32 check, or the fp64 referee's norm-wise rule AND its row rule.
@pytest.mark.parametrize("seed0,case,big", [(77, 671, False), (6, 23, True)])
def test_soak_regression_cases(dev, seed0, case, big):
    cam, g, tag, c = case_at(seed0, case, big)
    tol_worst = 1e-3 if c["hi"] >= 1.0 else ROW_TOL_WORST_P5000
    try:
        _check_against_oracle(cam, g, dev, seed=case, min_ok=0.98, tol_worst=tol_worst)
    except AssertionError as e:
        if not (str(e).startswith(("grad ", "oracle P=", "final_T")) or str(e) in ("depth", "colour")):
            raise
        note, rows = _adjudicate(cam, g, dev, case, tol_worst)
        assert not rows, (tag, rows, note)


def test_soak_regression_case_against_fp64(dev):
    """Seed 77 case 671 directly against the fp64 oracle: every gradient norm-wise inside 1e-4 (round 5: scales 1.17e-4)."""
    cam, g, tag, c = case_at(77, 671)
    kw = dict(colors_precomp=g.get("colors_precomp"), shs=g.get("shs"), scales=g.get("scales"), rotations=g.get("rotations"),
              cov3D_precomp=g.get("cov3D_precomp"), nthreads=4)
    o32 = TiledOracle(cam, g["means3D"], g["opacities"], **kw)
    o64 = TiledOracle(cam, g["means3D"], g["opacities"], f64=True, decisions_of=o32, **kw)
    dL = np.random.default_rng(671).uniform(-1, 1, (3, cam.image_height, cam.image_width)).astype(np.float32)
    dL[:, o32.ambiguous] = 0.0
    g64 = o64.backward(dL)
    _, _, _, grads, _ = _run_hip(cam, g, dev, dL=dL)
    for k, v in grads.items():
        assert rel_err(v, g64[k]) <= TOL, (tag, k, rel_err(v, g64[k]))


# The cases the soak stopped at or tallied in round 6 BEFORE preprocess_bwd's chain moved to fp64 (profiles/r06_fp64_chain.txt): every one an
# elongated Gaussian (a nearly singular conic) whose gradient row the fp32 chain put 1e-4 ... 6e-3 from fp64 -- in the HIP path and in the fp32
# oracle alike.  With the chain in fp64 the HIP path must sit well INSIDE the fp32 oracle's distance on them: a regression of the chain's
# precision (a float that crept back in, contraction, a reordered det) shows here, not in a bar that both fp32 evaluations miss together.
_CHAIN_CASES = [(9, 895, False), (9, 778, False), (2025, 234, False), (123, 602, False), (4242, 790, False), (3, 1255, False)]


@pytest.mark.parametrize("seed0,case,big", _CHAIN_CASES)
def test_fp64_chain_is_closer_to_fp64_than_the_fp32_oracle(dev, seed0, case, big):
    cam, g, tag, c = case_at(seed0, case, big)
    kw = dict(colors_precomp=g.get("colors_precomp"), shs=g.get("shs"), scales=g.get("scales"), rotations=g.get("rotations"),
              cov3D_precomp=g.get("cov3D_precomp"), nthreads=4)
    o32 = TiledOracle(cam, g["means3D"], g["opacities"], **kw)
    o64 = TiledOracle(cam, g["means3D"], g["opacities"], f64=True, decisions_of=o32, **kw)
    dL = np.random.default_rng(case).uniform(-1, 1, (3, cam.image_height, cam.image_width)).astype(np.float32)
    dL[:, o32.ambiguous] = 0.0
    g32, g64 = o32.backward(dL), o64.backward(dL)
    _, _, _, grads, _ = _run_hip(cam, g, dev, dL=dL)
    for k in ("means3D", "scales", "rotations"):
        e_h, e_o = rel_err(grads[k], g64[k]), rel_err(g32[k], g64[k])
        r_h, r_o = row_err(grads[k], g64[k])[0], row_err(g32[k], g64[k])[0]
        assert e_h <= TOL, (tag, k, e_h)
        assert e_h <= 0.75 * e_o + 1e-5, (tag, k, "norm-wise", e_h, e_o)      # measured: 0.03 ... 0.6 x the oracle's distance on these cases
        assert r_h <= 0.75 * r_o + 1e-4, (tag, k, "worst row", r_h, r_o)
