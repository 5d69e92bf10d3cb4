"""GPU box: gsr_fps_thin alone (1000 tracked points -> 100 farthest points + radius thinning, the first kernel of every rollout step), 200 calls.
GSR_HIP_LIB selects the library build (A/B on one box)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip
from gsdyn import synth_scene_params
from gsdyn import dynamics as D
dev = torch.device("cuda:0")
xyz = synth_scene_params(500_000, device=dev)["means3D"].detach()
pos = xyz[D.farthest_point_sampler(xyz[None], 1000)[0]].contiguous()
for _ in range(10):
    out = _hip.fps_thin_padded(pos, 100, 0.3, 0, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    out = _hip.fps_thin_padded(pos, 100, 0.3, 0, 0)
torch.cuda.synchronize()
print("%s: fps_thin %.1f us per call; kept %d; checksum %d %d" % (os.path.basename(_hip.LIB_PATH), (time.perf_counter() - t0) / 200 * 1e6, int(out[2]), int(out[0].sum()), int(out[1].sum())))
