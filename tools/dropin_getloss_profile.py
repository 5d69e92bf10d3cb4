"""GPU box: where the host time of the reference-shaped training step goes.  One step = gsdyn.get_loss for one camera
(two GaussianRasterizer calls through the unchanged API, torch loss glue) + loss.backward(), as bench.py's `separate_calls` times it
(/root/reference/src/tracking/train_utils.py:167-246, train_gs.py:25-39).  Prints: wall per camera, host issue time per camera (time until
the last launch is queued), GPU busy time of the library's kernels, then cProfile by own time and by cumulative time."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip
from gsdyn import LossWeights, get_loss, synth_ring_cameras, synth_scene_params, synth_targets
from gsdyn.dp import init_variables
from gsdyn.step import make_rigidity_variables
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
params = synth_scene_params(P, seed=0, device=dev)
cams = synth_ring_cameras(4, W, H, device=dev)
im_gt, seg_gt = synth_targets(W, H, device=dev)
variables = init_variables(P, dev)
variables.update(make_rigidity_variables(params, num_knn=20))
w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)
views = [dict(cam=c, im=im_gt, seg=seg_gt, id=i) for i, c in enumerate(cams)]
initial = len(sys.argv) > 1 and sys.argv[1] == "t0"

def step():
    for p in params.values():
        p.grad = None
    for d in views:
        loss, _ = get_loss(params, d, variables, initial, w)
        loss.backward()

for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
_hip.profile_begin()
for _ in range(5):
    step()
torch.cuda.synchronize()
prof = _hip.profile_end()
busy = sum(ms for ms, n in prof.values()) / 5 / len(views)
print(f"reference-shaped get_loss + backward, {'t = 0' if initial else 't > 0'}: wall {1e3 * t_all / N / len(views):.3f} ms per camera, host issue "
      f"{1e3 * t_issue / N / len(views):.3f} ms per camera, library kernels busy {busy:.3f} ms per camera")
print("library kernels per camera (us):", {k: round(1e3 * ms / 5 / len(views), 1) for k, (ms, n) in sorted(prof.items())})
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
for key, n in (("tottime", 28), ("cumulative", 28)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    txt = s.getvalue()
    print(f"---- cProfile by {key} ({N * len(views)} camera steps)")
    print("\n".join(l[:170] for l in txt.splitlines()[4:]))
if os.environ.get("TORCH_PROFILER") == "1":       # host time per torch op / autograd node (self CPU time), same steps
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as tp:
        for _ in range(N):
            step()
        torch.cuda.synchronize()
    print(f"---- torch.profiler, CPU self time ({N * len(views)} camera steps)")
    print(tp.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=70))
