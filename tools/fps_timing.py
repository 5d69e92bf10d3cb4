import os, sys, time, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gs-dynamics_amd"))
from gsdyn import dynamics as D
from gsdyn import synth_scene_params
dev = torch.device("cuda:0")
for P in (100_000, 500_000):
    xyz = synth_scene_params(P, device=dev)["means3D"].detach()
    for _ in range(2): D.farthest_point_sampler(xyz[None], 1000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): idx = D.farthest_point_sampler(xyz[None], 1000)
    torch.cuda.synchronize()
    print(f"FPS 1000 of {P}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms  checksum {int(idx.sum())}")
