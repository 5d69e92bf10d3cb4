"""GPU box: per-stage times of the one-launch propagation network (gsr_gnn_propagate) from the 100 MHz stamps workgroup 0 leaves in the
workspace -- graph of the configs[4] rollout (128 padded rows, 768 padded relations, width 512)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from gsdyn import dynamics as D
dev = torch.device("cuda:0")
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
           rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = D.DynamicsPredictor(cfg, device=dev).eval()
N, E = 128, 768
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    state_t = (torch.rand(N, 9, generator=g) * 0.4).to(dev)
    a = torch.zeros(N, 2, device=dev); a[:100, 0] = 1; a[100, 1] = 1
    gi = torch.zeros(N, 1, device=dev); gi[:100] = 1
    act = torch.zeros(N, 3, device=dev)
    recv = torch.sort(torch.randint(0, 101, (E,), generator=g))[0].to(dev)
    send = torch.randint(0, 101, (E,), generator=g).to(dev)
    for _ in range(5):
        model._propagate_fused(state_t, a, gi, act, recv, send)
    torch.cuda.synchronize()
    ws = next(iter(model._gnn_ws.values()))
    st = ws[-16 - 256:-16].view(torch.int64).cpu().tolist()
    names = ["stage 1 (first layers, segments)", "encoder layer 2", "encoder layer 3", "invariant parts"]
    for s in range(3):
        names += [f"step {s}: effect @ W2|W3", f"step {s}: aggregate", f"step {s}: particle propagator"]
    names += ["head 1", "head 2", "head 3 + outputs"]
    for i, nm in enumerate(names):
        print("%-36s %7.1f us" % (nm, (st[i + 1] - st[i]) / 100.0))
    print("%-36s %7.1f us" % ("total", (st[len(names)] - st[0]) / 100.0))
