"""How many tile-list entries does the forward blend at all?  Reads the per-entry contribution bytes (BinningState::contrib, bit w = some
pixel of quad w blended the entry) that the forward leaves for the backward, for one view of the benchmark scene (100k Gaussians, 800^2)
and of the 500k / 1080p frame.  Usage (GPU box): python tools/contrib_stats.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "gs-dynamics_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params

def stats(P, W, H, **kw):
    dev = torch.device("cuda:0")
    params = synth_scene_params(P, device=dev, **kw)
    cam = synth_ring_cameras(8, W, H, device=dev)[0]
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
    C = dgr._C
    out = C.rasterize_gaussians(cam.bg, rv["means3D"], rv["colors_precomp"], rv["opacities"], rv["scales"], rv["rotations"], cam.scale_modifier,
                                torch.empty(0, device=dev), cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, cam.image_height,
                                cam.image_width, torch.empty(0, device=dev), cam.sh_degree, cam.campos, cam.prefiltered)
    D, binning = out[0], out[5]
    torch.cuda.synchronize()
    a = (D + 255) // 256 * 256
    c = binning[binning.numel() - a:][:D].cpu().numpy()
    image = out[6].cpu().numpy()
    al = lambda x: (x + 255) // 256 * 256
    N, T = H * W, ((H + 15) // 16) * ((W + 15) // 16)
    ranges = image[2 * al(4 * N):][:8 * T].view(np.uint32).reshape(T, 2)
    below = useless_below = 0
    for lo, hi in ranges:
        nz = np.flatnonzero(c[lo:hi])
        if nz.size:
            below += nz[-1] + 1
            useless_below += nz[-1] + 1 - nz.size
    print(f"   entries below their tile's deepest blended entry (what the backward walks): {below} = {below / D:.3f} of all; of those NOT blended "
          f"by any quad: {useless_below / max(below, 1):.3f}")
    # quad imbalance: the backward replays a tile in batches; every batch lasts as long as its LONGEST per-quad list.  Visits wasted by
    # the lockstep = sum over batches of (4 x max_q - sum_q) visits; "whole tile" = the same with one batch per tile (what a ring that
    # covers the whole tile -- the producer / consumer form with unbounded LDS -- would still pay: a wave owns a quad)
    for BB in (80, 128):
        tot = waste = tot_tile = waste_tile = 0
        for lo, hi in ranges:
            nz = np.flatnonzero(c[lo:hi])
            if not nz.size:
                continue
            walked = c[lo:lo + nz[-1] + 1][::-1]                    # deepest first, as the backward walks
            bits = np.unpackbits(walked[:, None], axis=1)[:, 4:]    # [n, 4] (bit order irrelevant for counts)
            nb = (len(walked) + BB - 1) // BB
            pad = np.zeros((nb * BB, 4), np.int64); pad[:len(walked)] = bits
            per = pad.reshape(nb, BB, 4).sum(1)                     # visits per batch and quad
            tot += per.sum(); waste += (4 * per.max(1) - per.sum(1)).sum()
            pt = bits.sum(0)
            tot_tile += pt.sum(); waste_tile += 4 * pt.max() - pt.sum()
        print(f"   quad imbalance, batches of {BB}: visits {tot}, lockstep slots {tot + waste} (x{(tot + waste) / tot:.3f}); one batch per tile: x{(tot_tile + waste_tile) / tot_tile:.3f}")
    al256 = lambda x: (x + 255) // 256 * 256
    pl_off = 2 * al256(4 * D) + 2 * al256(8 * D)
    pl = binning[pl_off:pl_off + 4 * D].view(torch.int32).cpu().numpy()
    used = np.zeros(P, dtype=bool)
    np.logical_or.at(used, pl[c != 0], True)
    cnt_g = np.bincount(pl, minlength=P)
    print(f"   Gaussians with list entries: {int((cnt_g > 0).sum())}; of them blended somewhere: {int(used.sum())}; entries that belong to Gaussians "
          f"blended NOWHERE: {cnt_g[~used].sum() / D:.3f} of all entries")
    pop = np.unpackbits(c[:, None], axis=1)[:, 4:].sum(1)
    print(f"P={P} {W}x{H}: D={D} entries; no quad blended it: {np.mean(c == 0):.3f}; quads per entry (of entries with any): "
          + " ".join(f"{k}:{np.mean(pop[c != 0] == k):.3f}" for k in (1, 2, 3, 4)) + f"; mean quads per entry {pop.mean():.3f}")

if __name__ == "__main__":
    stats(100_000, 800, 800)
    stats(500_000, 1920, 1080)
