"""GPU box: how well does the longest-first ticket order (sorted by LIST LENGTH) balance the backward blend, whose work per tile is its
contributing visits (known from the forward's contribution bytes)?  For the 4- and 8-view benchmark step: per tile the list length n and the
lockstep wave slots (sum over 80-entry batches of the longest per-quad list); then greedy list scheduling on M identical workgroup slots
(M = 1536: six per CU) in ticket order by n (what the kernel does), by true work, and the lower bounds (total / M, longest job).
The processor-sharing model below charges a waiting quad's issue slots to its workgroup (the hardware gives them to the other waves of
the SIMD), so its absolute times run ~1.2x high; it is used only to COMPARE ticket orders: ordering by the true work instead of the list
length does not shorten the launch (within +-2 %), so the forward does not record a work estimate for the backward's queue."""
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


def tile_jobs(st):
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
        slots = 0
        if ml > 0:
            bits = pop[rg[t, 0]:rg[t, 0] + ml][::-1]
            nbt = (ml + 79) // 80
            padb = np.zeros((nbt * 80, 4), np.int64); padb[:ml] = bits
            slots = int(padb.reshape(nbt, 80, 4).sum(1).max(1).sum())
        jobs.append((n, ml, slots + 2))          # + a small constant per tile (head)
    return jobs


def makespan(jobs, key, M):
    order = sorted(jobs, key=key, reverse=True)
    heap = [0.0] * M
    for j in order:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + j[2])
    return max(heap)


# Processor-sharing model of the hardware: 256 CUs, K workgroup slots each; a CU with k resident workgroups retires wave slots at
# 1 / (k * t(k)) per workgroup, t(k) = ns per visit per SIMD with k waves on it (profiles/r05_visit_peak.txt, interpolated).
T_OF_K = {1: 252.0, 2: 165.0, 3: 124.0, 4: 112.0, 5: 106.0, 6: 103.3, 7: 100.0, 8: 98.0}


def share_sim(jobs, key, K, dt=0.25, head_ns=1500.0):
    order = sorted(jobs, key=key, reverse=True) if key else list(jobs)
    work = np.array([j[2] for j in order], np.float64)
    rem = np.zeros((256, K)); nxt = 0
    for c in range(256 * K):          # implicit first tickets: workgroup b on CU b % 256 (round-robin placement)
        if nxt < len(work):
            rem[c % 256, c // 256] = work[nxt] + head_ns / T_OF_K[K] / K; nxt += 1
    t = 0.0
    tk = np.array([0.0] + [T_OF_K[k] for k in range(1, K + 1)])
    while True:
        busy = rem > 0
        k = busy.sum(1)
        if not k.any():
            return t
        rate = np.where(k > 0, dt * 1e3 / (np.maximum(k, 1) * np.maximum(tk[k], 1.0)), 0.0)       # wave slots per workgroup per step
        rem = np.where(busy, rem - rate[:, None], rem)
        done = busy & (rem <= 0)
        for c, sl in zip(*np.nonzero(done)):
            if nxt < len(work):
                rem[c, sl] = work[nxt] + head_ns / T_OF_K[K] / K; nxt += 1
            else:
                rem[c, sl] = 0.0
        t += dt


for V in (1, 4, 8):
    cams = synth_ring_cameras(max(V, 4), S, S, device=dev)[:V]
    out = _hip.rasterize_forward_batch(list(cams), rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"], rv["rotations"], None,
                                       prepare_backward=True)
    torch.cuda.synchronize()
    jobs = [j for st in out[3] for j in tile_jobs(st)]
    tot = sum(j[2] for j in jobs)
    for M in (1024, 1536):
        lb = max(tot / M, max(j[2] for j in jobs))
        print(f"V={V} busy tiles {len(jobs)}, total slots {tot}, M={M}: lower bound {lb:.0f}; order by list length {makespan(jobs, lambda j: j[0], M):.0f} "
              f"(x{makespan(jobs, lambda j: j[0], M) / lb:.3f}); by walked length {makespan(jobs, lambda j: j[1], M):.0f} (x{makespan(jobs, lambda j: j[1], M) / lb:.3f}); "
              f"by true work {makespan(jobs, lambda j: j[2], M):.0f} (x{makespan(jobs, lambda j: j[2], M) / lb:.3f}); longest job {max(j[2] for j in jobs)}")
    for K in (4, 6, 8):
        ideal = tot * T_OF_K[K] / 256 / 1e3      # a wave slot occupies all four SIMDs of its CU
        print(f"     processor-sharing model, {K} workgroups per CU: all SIMDs full to the end {ideal:.1f} us; tickets by list length {share_sim(jobs, lambda j: j[0], K):.1f} us, "
              f"by walked length {share_sim(jobs, lambda j: j[1], K):.1f}, by true work {share_sim(jobs, lambda j: j[2], K):.1f}")
    n = np.array([j[0] for j in jobs]); w = np.array([j[2] for j in jobs])
    print(f"     correlation of work with list length {np.corrcoef(n, w)[0, 1]:.3f}; work / n: mean {np.mean(w / n):.3f}, p10 {np.percentile(w / n, 10):.3f}, p90 {np.percentile(w / n, 90):.3f}")
