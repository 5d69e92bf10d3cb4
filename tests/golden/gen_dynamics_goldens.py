"""Generate ``dynamics_host.npz``: golden vectors for SURVEY.md section 8f row N4 (graph building, GNN rollout step, motion
interpolation), captured by IMPORTING the Python reference in the build container.  Run once, here:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_dynamics_goldens.py

What is pinned:
  * ``construct_edges_from_states``   /root/reference/src/data/dataset.py:88-147   (seeded 11+1-node states, topk, both connect_all)
  * ``DynamicsPredictor.forward``     /root/reference/src/gnn/model.py:70-246      (rope config dims at width 32, seed-0 weights)
  * ``interpolate_motions``, ``mat2quat``, ``quat2mat``, ``relations_to_matrix``
                                      /root/reference/src/render/utils.py:50-243
  * ``fps_rad_idx_torch``             /root/reference/src/data/utils.py:50-65      (with its random start index fixed)
``dgl.geometry.farthest_point_sampler`` is an absent third-party function: NOT pinned.
Only inputs, outputs and randomly initialised weights are stored -- data, no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dynamics_host.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("open3d", "cv2", "dgl", "dgl.geometry", "ipdb", "tqdm"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["dgl.geometry"].farthest_point_sampler = None
    sys.modules["dgl"].geometry = sys.modules["dgl.geometry"]
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    sys.path.insert(0, REF)
    from gnn.model import DynamicsPredictor            # noqa
    from data.dataset import construct_edges_from_states  # noqa
    from data.utils import fps_rad_idx_torch           # noqa
    from render import utils as rutils                 # noqa
    return DynamicsPredictor, construct_edges_from_states, fps_rad_idx_torch, rutils


def main():
    DynamicsPredictor, construct_edges, fps_rad_idx_torch, rutils = import_reference()
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- edges: 11 object particles along a noisy curve + 1 tool particle
    N = 11
    t = torch.linspace(0, 1, N)
    obj = torch.stack([0.3 * t, 0.05 * torch.sin(6 * t), 0.02 * torch.rand(N, generator=g)], 1)
    tool = torch.tensor([[0.12, 0.05, 0.01]])
    states = torch.cat([obj, tool]).float()
    mask = torch.ones(N + 1, dtype=torch.bool)
    tool_mask = torch.zeros(N + 1, dtype=torch.bool); tool_mask[N] = True
    out["edge_states"] = states.numpy()
    for name, kw in (("a", dict(adj_thresh=0.08, topk=5, connect_all=False)), ("b", dict(adj_thresh=0.06, topk=3, connect_all=True))):
        Rr, Rs = construct_edges(states.clone(), kw["adj_thresh"], mask=mask, tool_mask=tool_mask, topk=kw["topk"], connect_all=kw["connect_all"])
        out[f"edge_{name}_Rr"], out[f"edge_{name}_Rs"] = Rr.numpy(), Rs.numpy()
        out[f"edge_{name}_cfg"] = np.array([kw["adj_thresh"], kw["topk"], float(kw["connect_all"])], np.float32)
        out[f"edge_{name}_rel"] = rutils.relations_to_matrix(Rr[None], Rs[None]).numpy()

    # ---- DynamicsPredictor forward: rope config dims, width 32, seed-0 weights
    cfg = dict(verbose=False, nf_particle=32, nf_relation=32, nf_effect=32, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
               rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
    torch.manual_seed(0)
    model = DynamicsPredictor(dict(cfg), "cpu").eval()
    for k, v in model.state_dict().items():
        out["gnn_w_" + k] = v.numpy().copy()
    out["gnn_cfg_keys"] = np.array(sorted(k for k in cfg if k != "verbose"))
    out["gnn_cfg_vals"] = np.array([cfg[k] for k in sorted(k for k in cfg if k != "verbose")], np.int64)
    n_his = cfg["n_his"]
    hist = torch.stack([states + 0.002 * i for i in range(n_his)])[None]          # [1,n_his,N+1,3]
    action = torch.zeros(1, N + 1, 3); action[0, N] = torch.tensor([0.01, -0.004, 0.002])
    attrs = torch.zeros(1, N + 1, 2); attrs[0, :N, 0] = 1; attrs[0, N:, 1] = 1
    p_instance = torch.ones(1, N, 1)
    Rr, Rs = torch.tensor(out["edge_a_Rr"])[None], torch.tensor(out["edge_a_Rs"])[None]
    with torch.no_grad():
        pred_pos, pred_motion = model(state=hist, attrs=attrs, Rr=Rr, Rs=Rs, p_instance=p_instance, action=action)
    out["gnn_state"], out["gnn_action"], out["gnn_attrs"] = hist.numpy(), action.numpy(), attrs.numpy()
    out["gnn_pred_pos"], out["gnn_pred_motion"] = pred_pos.numpy(), pred_motion.numpy()

    # ---- quaternion helpers
    q = torch.randn(64, 4, generator=g)
    R = rutils.quat2mat(q)
    out["q_in"], out["q_mat"], out["q_back"] = q.numpy(), R.numpy(), rutils.mat2quat(R).numpy()

    # ---- interpolate_motions: bones = the object particles, small rigid-ish motion, 300 particles around them
    bones = obj.clone()
    ang = 0.2
    Rz = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
    motions = (bones - bones.mean(0)) @ Rz.T + bones.mean(0) - bones + torch.tensor([0.01, 0.0, 0.005]) + 0.001 * torch.randn(N, 3, generator=g)
    rel = torch.tensor(out["edge_a_rel"])[:N, :N]
    xyz = bones[torch.randint(0, N, (300,), generator=g)] + 0.02 * torch.randn(300, 3, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(300, 4, generator=g), dim=-1)
    xyz_new, rot_new, weights = rutils.interpolate_motions(bones=bones, motions=motions, relations=rel, xyz=xyz, quat=quat, device="cpu")
    out["im_bones"], out["im_motions"], out["im_rel"], out["im_xyz"], out["im_quat"] = (
        bones.numpy(), motions.numpy(), rel.numpy(), xyz.numpy(), quat.numpy())
    out["im_xyz_new"], out["im_rot_new"], out["im_weights"] = xyz_new.numpy(), rot_new.numpy(), weights.numpy()

    # ---- radius FPS with the random start index fixed to 3
    pts = torch.rand(200, 3, generator=g) * torch.tensor([0.3, 0.1, 0.05])
    import data.utils as du
    orig = np.random.randint
    np.random.randint = lambda *a, **k: 3
    try:
        sel, idx = fps_rad_idx_torch(pts, 0.03)
    finally:
        np.random.randint = orig
    out["fpsr_pts"], out["fpsr_idx"], out["fpsr_sel"] = pts.numpy(), idx.numpy(), sel.numpy()

    # ---- BASELINE.json configs[0]: the rope demo cloud (assets/demo/pcd.ply, 3 491 points), 100 bones, ONE GNN step at the
    # rope.yaml width (512) with seed-0 weights.  The weights are NOT stored (14 MB): a module that creates its Linear layers
    # in the same order draws the same initial values from torch.manual_seed(0).
    raw = open("/root/reference/assets/demo/pcd.ply", "rb").read()
    body = raw[raw.index(b"end_header\n") + len(b"end_header\n"):]
    rec = np.frombuffer(body, dtype=np.dtype([("xyz", "<f8", 3), ("rgb", "u1", 3)]), count=3491)
    cloud = torch.tensor(rec["xyz"].astype(np.float32))
    pick = [0]                                            # farthest point sampling, first maximum (DGL is absent)
    mind = torch.full((cloud.shape[0],), float("inf"))
    for _ in range(99):
        mind = torch.minimum(mind, ((cloud - cloud[pick[-1]]) ** 2).sum(-1))
        pick.append(int(torch.argmax(mind)))
    bones = cloud[torch.tensor(pick)]
    eef = bones.mean(0, keepdim=True) + torch.tensor([[0.0, 0.0, 0.05]])
    st = torch.cat([bones, eef])
    Nn = st.shape[0]
    m1 = torch.ones(Nn, dtype=torch.bool); tm = torch.zeros(Nn, dtype=torch.bool); tm[-1] = True
    Rr1, Rs1 = construct_edges(st.clone(), 0.08, mask=m1, tool_mask=tm, topk=5, connect_all=False)
    cfg1 = dict(cfg, nf_particle=512, nf_relation=512, nf_effect=512)
    torch.manual_seed(0)
    model1 = DynamicsPredictor(dict(cfg1), "cpu").eval()
    hist1 = st[None, None].repeat(1, n_his, 1, 1)
    act1 = torch.zeros(1, Nn, 3); act1[0, -1] = torch.tensor([0.01, 0.0, -0.005])
    at1 = torch.zeros(1, Nn, 2); at1[0, :-1, 0] = 1; at1[0, -1, 1] = 1
    with torch.no_grad():
        pp1, pm1 = model1(state=hist1, attrs=at1, Rr=Rr1[None], Rs=Rs1[None], p_instance=torch.ones(1, Nn - 1, 1), action=act1)
    out["cfg1_cloud"], out["cfg1_pick"], out["cfg1_eef"], out["cfg1_action"] = cloud.numpy(), np.array(pick), eef.numpy(), act1.numpy()
    out["cfg1_n_rel"] = np.array([Rr1.shape[0]])
    out["cfg1_pred_pos"], out["cfg1_pred_motion"] = pp1.numpy(), pm1.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
