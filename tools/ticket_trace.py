"""GPU box, trace build (tools/build_variant_all.sh trace -DGSR_TRACE_TICKETS; GSR_HIP_LIB=.../libgsr_trace.so):
timeline of the backward blend's tickets -- when each workgroup started and finished each tile -- to see where a launch's
time goes (start-up, steady state, tail).  V=<views> env.  Times in us since workgroup 0 entered the kernel."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
V = int(os.environ.get("V", "1"))
params = synth_scene_params(100_000, device=dev)
cams = synth_ring_cameras(max(V, 4), 800, 800, device=dev)[:V]
dL = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (V, 3, 800, 800)).astype(np.float32), device=dev)
lib = _hip.load_library()
for _ in range(5):
    render_step_views(params, cams, dL)
torch.cuda.synchronize()
n = 65536
buf = (C.c_uint32 * (4 * n))()
lib.gsr_debug_ticket_trace.argtypes = [C.c_void_p, C.c_int]
lib.gsr_debug_ticket_trace(buf, n)
a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 4).astype(np.int64)
a = a[a[:, 3] > 0]
if os.environ.get("TRACE_DUMP"):
    np.save(os.environ["TRACE_DUMP"], a)
# one launch's tickets: the trace holds the last launch (same ticket ids overwrite)
base = a[:, 0].min()
t0, t1, wg, ln = (a[:, 0] - base) / 100.0, (a[:, 1] - base) / 100.0, a[:, 2], a[:, 3]
print(f"V={V} tickets {len(a)}  workgroups {len(np.unique(wg))}  entries {ln.sum()}  list length min/median/max {ln.min()}/{int(np.median(ln))}/{ln.max()}")
print(f"first ticket start: min {t0.min():.1f} us  p50 of implicit tickets {np.median(t0[:len(np.unique(wg))]):.1f}  max of implicit {t0[:len(np.unique(wg))].max():.1f}")
print(f"last ticket end {t1.max():.1f} us")
dur = t1 - t0
print(f"ticket duration us: min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f};  ns per entry median {1e3 * np.median(dur / ln):.1f}")
edges = np.arange(0, t1.max() + 5, 5.0)
print("time window (us): workgroups busy (avg), tickets finishing, entries/us retired")
for lo in edges[:-1]:
    hi = lo + 5
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / 5.0
    fin = ((t1 >= lo) & (t1 < hi))
    print(f"  {lo:6.0f}-{hi:<6.0f} {ov:7.1f} {fin.sum():6d} {ln[fin].sum() / 5.0:9.0f}")
# per-workgroup gaps between tickets
order = np.lexsort((t0, wg))
g = []
for i, j in zip(order[:-1], order[1:]):
    if wg[i] == wg[j]:
        g.append(t0[j] - t1[i])
if g:
    g = np.array(g)
    print(f"gap between a workgroup's tickets us: median {np.median(g):.2f} p90 {np.percentile(g, 90):.2f}")
