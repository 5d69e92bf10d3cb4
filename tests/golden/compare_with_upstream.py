"""Pin parity against the REAL upstream extension -- for a machine that has it (this repo's containers do not).

    # on a CUDA box with JonathonLuiten/diff-gaussian-rasterization-w-depth installed (gs-dynamics README.md:28-32):
    python tests/golden/compare_with_upstream.py [--device cuda]

Feeds the inputs of ``raster_cases.npz`` (and the per-view inputs of ``raster_cases_views.npz``) through upstream's
``GaussianRasterizer`` exactly as gs-dynamics calls it (/root/reference/src/tracking/train_utils.py:174-192: keyword arguments
means3D / means2D / opacities / colors_precomp / scales / rotations, settings built like tracking/helpers.py:20-32) and diffs
colour, depth, radii and every input gradient against the stored vectors, which are the outputs of this repo's oracle
(oracle/gsr_oracle.c).  Until this script has been run somewhere, parity of the oracle -- and therefore of the HIP path that is
tested against it -- is UNPINNED: the conventions recalled in SURVEY.md Appendix A (A-1 straight-through clamp, A-3 clamp mask,
A-4 depth without gradient and without alpha normalisation, det^2 + 1e-7) are exactly what a mismatch here would expose.
Exit code 0 = every case within 1e-4 (the north star's bar), 1 = mismatch (the table says where), 2 = upstream not importable.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    try:
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    except Exception as e:  # noqa: BLE001
        print("upstream diff_gaussian_rasterization is not importable here:", e)
        return 2
    if "gs-dynamics_amd" in (getattr(sys.modules["diff_gaussian_rasterization"], "__file__", "") or ""):
        print("this is the MI355X drop-in, not upstream: run on a box with the CUDA extension and without gs-dynamics_amd on sys.path")
        return 2
    dev = torch.device(args.device)
    t = lambda a, **k: torch.tensor(np.asarray(a, np.float32), device=dev, **k)  # noqa: E731
    bad = 0

    def run(cam, g, colours, dL):
        H, W = int(cam[0]), int(cam[1])
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(cam[2]), tanfovy=float(cam[3]), bg=t(cam[4:7]), scale_modifier=1.0,
            viewmatrix=t(cam[7:23]).reshape(1, 4, 4), projmatrix=t(cam[23:39]).reshape(1, 4, 4), sh_degree=0,
            campos=t(cam[39:42]), prefiltered=False)
        x = {k: t(g[k], requires_grad=True) for k in ("means3D", "scales", "rotations", "opacities")}
        col = t(colours, requires_grad=True)
        m2 = torch.zeros_like(x["means3D"], requires_grad=True)
        im, radii, depth = GaussianRasterizer(raster_settings=rs)(
            means3D=x["means3D"], means2D=m2, opacities=x["opacities"], colors_precomp=col, scales=x["scales"], rotations=x["rotations"])
        im.backward(gradient=t(dL))
        grads = {k: v.grad.cpu().numpy() for k, v in x.items()}
        grads["colors_precomp"], grads["means2D"] = col.grad.cpu().numpy(), m2.grad.cpu().numpy()
        return im.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), grads

    def report(tag, rows):
        nonlocal bad
        for what, e in rows:
            flag = "" if e <= TOL else "   <-- MISMATCH"
            bad += e > TOL
            print(f"{tag:28s} {what:22s} {e:.3e}{flag}")

    z = np.load(os.path.join(HERE, "raster_cases.npz"))
    for n in [str(x) for x in z["names"]]:
        g = {k: z[f"{n}/in_{k}"] for k in ("means3D", "scales", "rotations", "opacities")}
        im, radii, depth, grads = run(z[f"{n}/cam"], g, z[f"{n}/in_colors_precomp"], z[f"{n}/dL_dcolor"])
        ok = ~z[f"{n}/ambiguous"]
        rows = [("radii (count differing)", float((radii != z[f"{n}/radii"]).sum())),
                ("colour", float(np.abs(im - z[f"{n}/color"])[:, ok].max())),
                ("depth", rel(depth[:, ok], z[f"{n}/depth"][:, ok]))]
        rows += [("grad " + k, rel(grads[k], z[f"{n}/grad_{k}"])) for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations")]
        report(n, rows)
    z = np.load(os.path.join(HERE, "raster_cases_views.npz"))
    for n in [str(x) for x in z["names"]]:
        g = {k: z[f"{n}/in_{k}"] for k in ("means3D", "scales", "rotations", "opacities")}
        sums = {k: 0.0 for k in ("means3D", "opacities", "scales", "rotations")}
        for vi in range(len(z[f"{n}/view_cam"])):
            im, radii, depth, grads = run(z[f"{n}/cam"][vi], g, z[f"{n}/in_colours"][z[f"{n}/view_colour"][vi]], z[f"{n}/dL_dcolor"][vi])
            ok = ~z[f"{n}/ambiguous"][vi]
            report(f"{n}[{vi}]", [("radii (count differing)", float((radii != z[f"{n}/radii"][vi]).sum())),
                                  ("colour", float(np.abs(im - z[f"{n}/color"][vi])[:, ok].max())),
                                  ("depth", rel(depth[:, ok], z[f"{n}/depth"][vi][:, ok])),
                                  ("grad means2D", rel(grads["means2D"], z[f"{n}/grad_means2D"][vi])),
                                  ("grad colours", rel(grads["colors_precomp"], z[f"{n}/grad_colours_per_view"][vi]))])
            for k in sums:
                sums[k] = sums[k] + grads[k].astype(np.float64)
        report(n, [("sum over views: grad " + k, rel(sums[k], z[f"{n}/grad_sum_{k}"])) for k in sums])
    print("PARITY PINNED: every vector within 1e-4 of upstream" if not bad else f"{bad} quantities differ from upstream by more than 1e-4")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
