"""GPU box: would a FINER replay granularity shorten the backward blend?  Today a wave owns an 8x8 quad and a visit = (quad, entry): all 64
lanes evaluate the entry, a 64-lane reduction sums the nine partials.  Alternative: the wave's four DPP rows own the quad's four 4x4
blocks and each row walks ITS OWN compacted list (four different entries per trip, 16-lane reductions that never leave a DPP row).  A
trip then costs about what a visit costs now, and the number of trips is max over the rows of their list lengths.  This tool counts,
on the benchmark scene (100 k Gaussians, 800^2, one view), from the forward's own outputs re-evaluated per pixel:
  visits64   (quad, entry) pairs with a blending pixel        -> lockstep slots today (batches of 80, max over the four quads)
  visits16   (4x4 block, entry) pairs with a blending pixel   -> trips of the row scheme (max over a wave's four rows, then over the quads)
Model only -- it decides whether the restructuring is worth building."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")


def run(P, W, H, BB=80, **kw):
    params = synth_scene_params(P, seed=0, device=dev, **kw)
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
    cam = synth_ring_cameras(4, W, H, device=dev)[0]
    out = _hip.rasterize_forward_batch([cam], rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"], rv["rotations"], None,
                                       prepare_backward=True)
    torch.cuda.synchronize()
    st = out[3][0]
    v = _hip.debug_views(st)
    rec, pl, rg, nc = v["rec"], v["point_list"].long(), v["ranges"].cpu().numpy(), v["n_contrib"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ly, lx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
    quad = ((ly // 8) * 2 + (lx // 8)).reshape(-1)                       # 0..3
    blk = (((ly % 8) // 4) * 2 + ((lx % 8) // 4)).reshape(-1)            # 4x4 block inside the quad, 0..3
    oh_q = torch.nn.functional.one_hot(quad, 4).float()                  # [256, 4]
    oh_b = torch.nn.functional.one_hot(quad * 4 + blk, 16).float()       # [256, 16]
    oh_r = torch.nn.functional.one_hot((ly.reshape(-1) // 2), 8).float() # alternative: 16 lanes = two pixel rows of the quad? (8 x 2 strips) -- not used
    tot = dict(v64=0, slots64=0, v16=0, trips16=0, trips16_wave=0, pix=0, v64_tile=0, slots64_tile=0, trips16_tile=0, px_per_v64=0)
    hist = np.zeros(5, np.int64)
    for t in range(gx * gy):
        lo, hi = int(rg[t, 0]), int(rg[t, 1])
        if hi <= lo:
            continue
        tx, ty = t % gx, t // gx
        px = (tx * 16 + lx).reshape(-1).float(); py = (ty * 16 + ly).reshape(-1).float()
        inside = ((tx * 16 + lx) < W).reshape(-1) & ((ty * 16 + ly) < H).reshape(-1)
        last = torch.zeros(256, dtype=torch.long, device=dev)
        yy, xx = (ty * 16 + ly).reshape(-1).clamp(max=H - 1), (tx * 16 + lx).reshape(-1).clamp(max=W - 1)
        last = torch.where(inside, nc[yy, xx].long(), torch.zeros_like(yy))
        ml = int(last.max())
        if ml == 0:
            continue
        g = pl[lo:lo + ml]
        r = rec[g]                                              # [ml, 16]
        dx = r[:, 0:1] - px[None]; dy = r[:, 1:2] - py[None]
        power = -0.5 * (r[:, 2:3] * dx * dx + r[:, 4:5] * dy * dy) - r[:, 3:4] * dx * dy
        hit = (power <= 0) & (r[:, 5:6] * torch.exp(power) >= 1.0 / 255.0) & (torch.arange(ml, device=dev)[:, None] < last[None])
        hf = hit.float()
        a64 = (hf @ oh_q) > 0                                   # [ml, 4]
        a16 = ((hf @ oh_b) > 0).reshape(ml, 4, 4)               # [ml, quad, block]
        hist += np.bincount(a16.sum(2)[a64].cpu().numpy(), minlength=5)[:5]
        # back to front in batches of BB list positions
        a64r, a16r = a64.flip(0), a16.flip(0)
        nb = (ml + BB - 1) // BB
        p64 = torch.zeros((nb * BB, 4), device=dev); p64[:ml] = a64r.float()
        p16 = torch.zeros((nb * BB, 4, 4), device=dev); p16[:ml] = a16r.float()
        per64 = p64.reshape(nb, BB, 4).sum(1)                   # [nb, quad]
        per16 = p16.reshape(nb, BB, 4, 4).sum(1)                # [nb, quad, block]
        tot["v64"] += int(a64.sum()); tot["slots64"] += int(per64.max(1).values.sum())
        tot["v16"] += int(a16.sum()); tot["trips16"] += int(per16.max(2).values.max(1).values.sum())
        tot["trips16_wave"] += int(per16.max(2).values.sum())   # per wave, no lockstep between the quads (for the ratio of the two effects)
        tot["pix"] += int(hit.sum())
        tot["v64_tile"] += int(a64.sum(0).max()); tot["trips16_tile"] += int(a16.sum(0).max())
    print(f"P={P} {W}x{H}, batches of {BB}:")
    print(f"   pixel-entry pairs that blend {tot['pix']}; quad visits {tot['v64']} ({tot['pix'] / tot['v64']:.1f} of 64 pixels per visit); 4x4-block visits {tot['v16']} "
          f"({tot['pix'] / tot['v16']:.1f} of 16 pixels, {tot['v16'] / tot['v64']:.2f} blocks per quad visit; histogram of blocks per quad visit 1..4: "
          f"{(hist[1:] / hist[1:].sum()).round(3).tolist()})")
    print(f"   lockstep slots today (max over the quads per batch): {tot['slots64']} = {tot['slots64'] / tot['v64'] * 4:.3f} x visits / 4 ... {tot['slots64']} wave-trips per workgroup")
    print(f"   row scheme: trips per workgroup {tot['trips16']} = {tot['trips16'] / tot['slots64']:.3f} x today's slots (per wave without the lockstep between quads: "
          f"{tot['trips16_wave'] / 4:.0f}; ideal visits16 / 16 = {tot['v16'] / 16:.0f})")


run(100_000, 800, 800)
run(100_000, 800, 800, BB=128)
run(500_000, 1920, 1080)
