"""Generate ``rollout_host.npz``: the reference's autoregressive rollout (``DynamicsModule.rollout``,
/root/reference/src/render/dynamics_module.py:53-172) and the frame smoothing of ``collect_scene_data`` (:223-236), run by IMPORTING
the Python reference here on the CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_rollout_goldens.py

Stubs, all for absent third-party modules: ``dgl.geometry.farthest_point_sampler`` is replaced by a first-maximum farthest point
sampler (DGL's tie rule is therefore NOT pinned; everything downstream of the picks is), ``open3d`` / ``cv2`` / ``tqdm`` are empty
modules, and the random first index of ``fps_rad_idx_torch`` is fixed to 0.  ``interpolate_motions`` is called with device='cpu'
(its default is 'cuda').  The model is the reference's DynamicsPredictor at width 32 with torch.manual_seed(0) weights, rebuilt by
the test from the same seed.  Only inputs and outputs are stored.
"""
import functools
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rollout_host.npz")
CFG = dict(max_nobj=24, fps_radius=0.025, adj_thresh=0.07, topk=5, connect_all=False, dist_thresh=0.004)


def first_max_fps(pos, npoints, start_idx=0):
    out = torch.zeros((pos.shape[0], npoints), dtype=torch.long)
    for b in range(pos.shape[0]):
        p, cur = pos[b].float(), int(start_idx)
        mind = torch.full((p.shape[0],), float("inf"))
        for k in range(npoints):
            out[b, k] = cur
            mind = torch.minimum(mind, ((p - p[cur]) ** 2).sum(-1))
            cur = int(torch.argmax(mind))
    return out


def main():
    sys.dont_write_bytecode = True
    for name in ("open3d", "cv2", "dgl", "dgl.geometry", "ipdb", "tqdm"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["dgl.geometry"].farthest_point_sampler = first_max_fps
    sys.modules["dgl"].geometry = sys.modules["dgl.geometry"]
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    sys.path.insert(0, REF)
    from gnn.model import DynamicsPredictor
    from render import dynamics_module as dm
    from render import utils as rutils
    dm.interpolate_motions = functools.partial(rutils.interpolate_motions, device="cpu")
    np.random.randint = lambda *a, **k: 0

    g = torch.Generator().manual_seed(3)
    P, S = 1500, 7
    t = torch.rand(P, generator=g)
    centre = torch.stack([0.3 * t, 0.05 * torch.sin(6 * t), torch.zeros(P)], 1)
    xyz_0 = centre + 0.006 * torch.randn(P, 3, generator=g)
    rgb_0 = torch.rand(P, 3, generator=g)
    quat_0 = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=-1)
    opa_0 = torch.rand(P, 1, generator=g)
    inlier = np.sort(torch.randperm(P, generator=g)[:1300].numpy())
    steps = torch.tensor([[0, 0, 0], [0.01, 0.002, 0.0], [0.011, 0.002, 0.0], [0.02, 0.004, 0.003], [0.03, 0.0, 0.004],
                          [0.0305, 0.0, 0.004], [0.04, -0.004, 0.0]], dtype=torch.float32)
    eef_xyz = (torch.tensor([0.1, 0.03, 0.02]) + steps)[:, None, :]        # [S,1,3]; steps 2 and 5 move less than dist_thresh

    cfg = dict(verbose=False, nf_particle=32, nf_relation=32, nf_effect=32, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
               rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
    torch.manual_seed(0)
    model = DynamicsPredictor(dict(cfg), "cpu").eval()
    mod = dm.DynamicsModule.__new__(dm.DynamicsModule)
    mod.device, mod.model, mod.n_his = "cpu", model, 3
    mod.dist_thresh, mod.max_nobj, mod.adj_thresh = CFG["dist_thresh"], CFG["max_nobj"], CFG["adj_thresh"]
    mod.fps_radius, mod.topk, mod.connect_all = CFG["fps_radius"], CFG["topk"], CFG["connect_all"]
    xyz, rgb, quat, opa, bones, eef = mod.rollout(xyz_0, rgb_0, quat_0, opa_0, eef_xyz, S, inlier)
    out = dict(xyz_0=xyz_0.numpy(), rgb_0=rgb_0.numpy(), quat_0=quat_0.numpy(), opa_0=opa_0.numpy(), inlier=inlier, eef_xyz=eef_xyz.numpy(),
               cfg=np.array([CFG["max_nobj"], CFG["fps_radius"], CFG["adj_thresh"], CFG["topk"], float(CFG["connect_all"]), CFG["dist_thresh"]]),
               xyz=xyz.numpy(), quat=quat.numpy(), bones=bones.numpy(), eef=eef.numpy())
    # the smoothing block of collect_scene_data (:223-236), run on copies exactly as written there
    xs, rs, qs, os_, bs, es = (a.clone() for a in (xyz, rgb, quat, opa, bones, eef))
    cps = (xs - torch.concatenate([xs[0:1], xs[:-1]], dim=0)).norm(dim=-1).sum(dim=-1).nonzero().squeeze(1)
    cps = torch.cat([torch.tensor([0]), cps])
    for i in range(1, len(cps)):
        a, b = cps[i - 1], cps[i]
        if b - a < 2:
            continue
        for arr in (xs, rs, qs, os_, bs, es):
            arr[a:b] = torch.lerp(arr[a][None], arr[b][None], torch.linspace(0, 1, b - a + 1)[:, None, None])[:-1]
    qs = torch.nn.functional.normalize(qs, dim=-1)
    out.update(smooth_xyz=xs.numpy(), smooth_quat=qs.numpy(), smooth_bones=bs.numpy(), smooth_eef=es.numpy(), change_points=cps.numpy())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; change points", cps.tolist())


if __name__ == "__main__":
    main()
