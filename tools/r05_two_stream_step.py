"""GPU box: does splitting a step's views into two groups on two HIP streams pay?  (One group's latency-bound preprocess / binning / sort
chain would backfill the other group's blend-kernel drain.)  Times the 4- and 8-view fwd + bwd step of bench.py (gsdyn.step.render_step_views)
as ONE call per step against TWO concurrent calls (half the views each), alternating, with wall clock over many steps."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
P, S = 100_000, 800
params = synth_scene_params(P, seed=0, device=dev)
for V in (4, 8, 2):
    cams = synth_ring_cameras(max(V, 4), S, S, device=dev)[:V]
    dL = torch.tensor(np.random.default_rng(1234).uniform(-1, 1, (V, 3, S, S)).astype(np.float32), device=dev)
    h = V // 2
    groups = [(cams[:h], dL[:h].contiguous()), (cams[h:], dL[h:].contiguous())]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def one():
        render_step_views(params, cams, dL)

    def two():
        cur = torch.cuda.current_stream(dev)
        for st, (c, d) in zip(streams, groups):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                render_step_views(params, c, d)
        for st in streams:
            cur.wait_stream(st)

    def two_nosync():       # the same, and the host does not wait for group A's entry counts before it queues group B (counts checked after both)
        cur = torch.cuda.current_stream(dev)
        outs = []
        for st, (c, d) in zip(streams, groups):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(render_step_views(params, c, d, _defer_counts=True))
        for st in streams:
            cur.wait_stream(st)
        from diff_gaussian_rasterization import _hip
        for _ims, g in outs:
            assert _hip.forward_counts_ok(g["_states"])
        return outs

    def two_seq():          # the same two calls on ONE stream (what splitting alone costs)
        for c, d in groups:
            render_step_views(params, c, d)

    res = {}
    for rnd in range(3):
        for name, fn in (("one call", one), ("two streams", two), ("two streams, counts deferred", two_nosync), ("two calls, one stream", two_seq)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            N = 40
            t0 = time.perf_counter()
            for _ in range(N):
                fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(1e6 * (time.perf_counter() - t0) / N)
    print(f"V={V}: " + "; ".join(f"{k}: " + " / ".join(f"{x:.0f}" for x in v) + " us" for k, v in res.items()), flush=True)
