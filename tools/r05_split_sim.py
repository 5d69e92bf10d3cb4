"""GPU box: would SPLITTING long tiles shorten the backward blend?  A tile's list is a sequential chain per pixel, so today a ticket is a
whole tile and the launch lasts at least as long as its longest ticket (one view: 857 entries under ~4-way sharing = the whole 80 us).
With a checkpoint of (T, accumulated colour) per pixel every SEG list positions from the forward, the backward could replay the
segments of a tile as independent tickets.  This tool takes the real per-tile contribution bytes of the benchmark scene and runs the
processor-sharing model of tools/r05_lpt_sim.py on whole tiles and on segments (every segment pays the ticket head again), for one, two,
four and eight views.  Model only -- it decides whether the restructuring is worth building."""
import heapq, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
P, S = 100_000, 800
params = synth_scene_params(P, seed=0, device=dev)
with torch.no_grad():
    rv = {k: v.detach() for k, v in params2rendervar(params).items()}


def tile_jobs(st, seg):
    D, H, W = int(st.num_rendered), int(st.H), int(st.W)
    v = _hip.debug_views(st)
    rg = v["ranges"].cpu().numpy().astype(np.int64)
    nc = v["n_contrib"].cpu().numpy()
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), nc.dtype); pad[:H, :W] = nc
    max_last = pad.reshape(gy, 16, gx, 16).max((1, 3)).reshape(-1).astype(np.int64)
    al = lambda x: (x + 255) // 256 * 256
    nb = max(1, (D + 2047) // 2048)
    off = 2 * al(4 * D) + 2 * al(8 * D) + al(4 * D) + al(1024 * (nb + 1))
    c = st.binning[off:off + D].cpu().numpy()
    pop = np.unpackbits(c[:, None], axis=1)[:, 4:]
    jobs = []
    for t in range(rg.shape[0]):
        n, ml = int(rg[t, 1] - rg[t, 0]), int(max_last[t])
        if n <= 0:
            continue
        nseg = max(1, (n + seg - 1) // seg) if seg else 1
        for k in range(nseg):
            lo = k * seg if seg else 0
            hi = min(n, lo + seg) if seg else n
            w_hi = min(ml, hi)
            slots = 0
            if w_hi > lo:
                bits = pop[rg[t, 0] + lo:rg[t, 0] + w_hi][::-1]
                m = w_hi - lo
                nbt = (m + 79) // 80
                padb = np.zeros((nbt * 80, 4), np.int64); padb[:m] = bits
                slots = int(padb.reshape(nbt, 80, 4).sum(1).max(1).sum())
            jobs.append((hi - lo, w_hi - lo, slots + 2))
    return jobs


T_OF_K = {1: 252.0, 2: 165.0, 3: 124.0, 4: 112.0, 5: 106.0, 6: 103.3, 7: 100.0, 8: 98.0}


def share_sim(jobs, K, head_ns, dt=0.25):
    order = sorted(jobs, key=lambda j: j[0], reverse=True)
    work = np.array([j[2] for j in order], np.float64)
    rem = np.zeros((256, K)); nxt = 0
    for c in range(256 * K):
        if nxt < len(work):
            rem[c % 256, c // 256] = work[nxt] + head_ns / T_OF_K[K] / K; nxt += 1
    t = 0.0
    tk = np.array([0.0] + [T_OF_K[k] for k in range(1, K + 1)])
    while True:
        busy = rem > 0
        k = busy.sum(1)
        if not k.any():
            return t
        rate = np.where(k > 0, dt * 1e3 / (np.maximum(k, 1) * np.maximum(tk[k], 1.0)), 0.0)
        rem = np.where(busy, rem - rate[:, None], rem)
        done = busy & (rem <= 0)
        for c, sl in zip(*np.nonzero(done)):
            if nxt < len(work):
                rem[c, sl] = work[nxt] + head_ns / T_OF_K[K] / K; nxt += 1
            else:
                rem[c, sl] = 0.0
        t += dt


for V in (1, 2, 4, 8):
    cams = synth_ring_cameras(max(V, 4), S, S, device=dev)[:V]
    out = _hip.rasterize_forward_batch(list(cams), rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"], rv["rotations"], None,
                                       prepare_backward=True)
    torch.cuda.synchronize()
    for seg in (0, 512, 384, 256, 192):
        jobs = [j for st in out[3] for j in tile_jobs(st, seg)]
        tot = sum(j[2] for j in jobs)
        line = f"V={V} segment {seg or 'whole tile':>10}: tickets {len(jobs):5d}, wave slots {tot:7d}, longest {max(j[2] for j in jobs):4d};"
        for K in (4, 6):
            for head in (1500.0, 3000.0):
                line += f"  K={K} head {head / 1e3:.1f} us: {share_sim(jobs, K, head):6.1f} us"
        print(line, flush=True)
