"""GPU box: gradient of every view-independent t > 0 term at BASELINE size in three evaluations -- fused kernels (fp32), the
torch formulas in fp32, the torch formulas in fp64 -- to see which term (if any) is ill-conditioned in fp32."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import params2rendervar, synth_scene_params
from gsdyn.step import _SHARED_NAMES, _shared_terms, make_rigidity_variables
dev = torch.device("cuda:0")
torch.manual_seed(1234)
P = int(os.environ.get("P", 100_000))
params = synth_scene_params(P, device=dev)
rig = make_rigidity_variables(params, num_knn=20)
with torch.no_grad():
    params["means3D"].add_(0.003 * torch.randn_like(params["means3D"]))
    params["unnorm_rotations"].add_(0.02 * torch.randn_like(params["unnorm_rotations"]))
tv32 = {k: v for k, v in rig.items() if k not in ("rev_ptr", "rev_edge")}
tv64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in tv32.items()}
for name in list(_SHARED_NAMES) + ["all"]:
    w = {k: (1.0 if (k == name or name == "all") else 0.0) for k in _SHARED_NAMES}
    if name == "all":
        w = dict(rigid=200.0, rot=4.0, iso=1000.0, floor=2.0, bg=200.0)
    res = {}
    for tag, pp, vv in (("fused", params, rig), ("t32", params, tv32), ("t64", {k: v.detach().double() for k, v in params.items()}, tv64)):
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in pp.items()}
        rv = params2rendervar(leaf)
        m, r = rv["means3D"], rv["rotations"]
        r.retain_grad()
        tot, each = _shared_terms(leaf, rv, vv, w)
        tot.backward()
        res[tag] = (leaf["means3D"].grad.double(), r.grad.double(), leaf["unnorm_rotations"].grad.double(), float(tot))
    def rel(a, b):
        return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()
    print(f"{name:6s} value fused/t32/t64 {res['fused'][3]:.6e} {res['t32'][3]:.6e} {res['t64'][3]:.6e} | "
          + " | ".join(f"{lab}: fused-vs-t64 {rel(res['fused'][i], res['t64'][i]):.1e}, t32-vs-t64 {rel(res['t32'][i], res['t64'][i]):.1e}, fused-vs-t32 {rel(res['fused'][i], res['t32'][i]):.1e}"
                       for i, lab in ((0, "d means3D"), (1, "d rot"), (2, "d unnorm_rot"))))
