"""CPU tests of the rollout plumbing against golden vectors captured from the imported
reference (tests/golden/gen_dynamics_goldens.py -> dynamics_host.npz)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "dynamics_host.npz"))


def _edge_inputs(gold):
    states = torch.tensor(gold["edge_states"])
    N = states.shape[0]
    mask = torch.ones(N, dtype=torch.bool)
    tool = torch.zeros(N, dtype=torch.bool)
    tool[N - 1] = True
    return states, mask, tool


@pytest.mark.parametrize("case", ["a", "b"])
def test_relations_match_reference(gold, case):
    from gsdyn.dynamics import construct_edges, edges_to_dense, relations_to_matrix
    states, mask, tool = _edge_inputs(gold)
    thr, topk, call = gold[f"edge_{case}_cfg"]
    recv, send = construct_edges(states, float(thr), mask, tool, topk=int(topk), connect_all=bool(call))
    Rr, Rs = edges_to_dense(recv, send, states.shape[0])
    assert np.array_equal(Rr.numpy(), gold[f"edge_{case}_Rr"]) and np.array_equal(Rs.numpy(), gold[f"edge_{case}_Rs"])
    assert np.array_equal(relations_to_matrix(recv, send, states.shape[0]).numpy(), gold[f"edge_{case}_rel"])


def _model(gold):
    from gsdyn.dynamics import DynamicsPredictor
    cfg = {str(k): int(v) for k, v in zip(gold["gnn_cfg_keys"], gold["gnn_cfg_vals"])}
    model = DynamicsPredictor(cfg)
    sd = {k[len("gnn_w_"):]: torch.tensor(gold[k]) for k in gold.files if k.startswith("gnn_w_")}
    model.load_state_dict(sd, strict=True)      # the reference's parameter names load unchanged
    return model.eval()


def test_dynamics_predictor_matches_reference_dense_and_index_form(gold):
    from gsdyn.dynamics import construct_edges
    model = _model(gold)
    state, action, attrs = (torch.tensor(gold[k]) for k in ("gnn_state", "gnn_action", "gnn_attrs"))
    N = attrs.shape[1]
    p_instance = torch.ones(1, N - 1, 1)
    Rr, Rs = torch.tensor(gold["edge_a_Rr"])[None], torch.tensor(gold["edge_a_Rs"])[None]
    with torch.no_grad():
        pos_d, mot_d = model(state=state, attrs=attrs, p_instance=p_instance, action=action, Rr=Rr, Rs=Rs)
        states, mask, tool = _edge_inputs(gold)
        recv, send = construct_edges(states, 0.08, mask, tool, topk=5, connect_all=False)
        pos_i, mot_i = model(state=state, attrs=attrs, p_instance=p_instance, action=action, receivers=recv, senders=send)
    np.testing.assert_allclose(pos_d.numpy(), gold["gnn_pred_pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mot_d.numpy(), gold["gnn_pred_motion"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(pos_i.numpy(), gold["gnn_pred_pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mot_i.numpy(), gold["gnn_pred_motion"], rtol=1e-4, atol=1e-6)


def test_quaternion_helpers_match_reference(gold):
    from gsdyn.dynamics import mat2quat, quat2mat
    q = torch.tensor(gold["q_in"])
    R = quat2mat(q)
    np.testing.assert_allclose(R.numpy(), gold["q_mat"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mat2quat(torch.tensor(gold["q_mat"])).numpy(), gold["q_back"], rtol=1e-5, atol=1e-6)
    # all four branches of mat2quat: rotations by pi about x, y, z have trace -1
    for axis in range(3):
        Rm = -torch.eye(3)
        Rm[axis, axis] = 1.0
        qq = mat2quat(Rm[None])[0]              # un-normalised like the reference's (its callers normalise)
        expect = torch.zeros(4)
        expect[axis + 1] = 1.0
        np.testing.assert_allclose(torch.nn.functional.normalize(qq, dim=0).abs().numpy(), expect.numpy(), atol=1e-6)


def test_interpolate_motions_matches_reference(gold):
    from gsdyn.dynamics import interpolate_motions
    t = lambda k: torch.tensor(gold[k])  # noqa: E731
    xyz_new, rot_new, weights = interpolate_motions(t("im_bones"), t("im_motions"), t("im_rel"), t("im_xyz"), quat=t("im_quat"))
    np.testing.assert_allclose(weights.numpy(), gold["im_weights"], rtol=2e-4, atol=1e-7)   # the reference's cdist uses the mm form
    np.testing.assert_allclose(xyz_new.numpy(), gold["im_xyz_new"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(rot_new.numpy(), gold["im_rot_new"], rtol=1e-4, atol=2e-6)


def test_radius_fps_matches_reference_and_fps_properties(gold):
    from gsdyn.dynamics import farthest_point_sampler, fps_radius
    pts = torch.tensor(gold["fpsr_pts"])
    sel, idx = fps_radius(pts, 0.03, start_idx=3)
    assert np.array_equal(idx.numpy(), gold["fpsr_idx"])
    np.testing.assert_allclose(sel.numpy(), gold["fpsr_sel"])
    # farthest point sampling (DGL's routine is not pinned): start index first, no repeats, greedy max-min property
    pick = farthest_point_sampler(pts[None], 20, start_idx=5)[0]
    assert int(pick[0]) == 5 and len(set(pick.tolist())) == 20
    d = torch.cdist(pts, pts) ** 2
    mind = torch.full((pts.shape[0],), float("inf"))
    for k in range(19):
        mind = torch.minimum(mind, d[pick[k]])
        assert float(mind[pick[k + 1]]) == float(mind.max())


def test_rollout_step_runs_and_is_consistent(gold):
    from gsdyn.dynamics import rollout_step
    model = _model(gold)
    states, _, _ = _edge_inputs(gold)
    nobj = states.shape[0] - 1
    hist = torch.stack([states[:nobj] + 0.002 * i for i in range(3)])
    eef_hist = states[nobj:][None].repeat(3, 1, 1)
    g = torch.Generator().manual_seed(1)
    xyz = states[:nobj][torch.randint(0, nobj, (500,), generator=g)] + 0.01 * torch.randn(500, 3, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(500, 4, generator=g), dim=-1)
    pred, xyz_new, quat_new, (recv, send) = rollout_step(model, hist, eef_hist, states[nobj:] + 0.01, xyz, quat, 0.08, 5)
    assert pred.shape == (nobj, 3) and xyz_new.shape == xyz.shape and quat_new.shape == quat.shape
    assert torch.isfinite(xyz_new).all() and torch.allclose(quat_new.norm(dim=-1), torch.ones(500), atol=1e-5)
    assert recv.shape == send.shape and recv.numel() > nobj          # at least the self relations


def test_baseline_config1_rope_demo_step(gold):
    """BASELINE.json configs[0]: rope demo cloud -> 100 bones -> relations -> ONE DynamicsPredictor step at the rope.yaml
    width with seed-0 weights (same Linear creation order => same initial values as the reference's module)."""
    from gsdyn.dynamics import DynamicsPredictor, construct_edges, farthest_point_sampler
    cloud = torch.tensor(gold["cfg1_cloud"])
    pick = farthest_point_sampler(cloud[None], 100, start_idx=0)[0]
    assert np.array_equal(pick.numpy(), gold["cfg1_pick"])
    st = torch.cat([cloud[pick], torch.tensor(gold["cfg1_eef"])])
    N = st.shape[0]
    mask = torch.ones(N, dtype=torch.bool)
    tool = torch.zeros(N, dtype=torch.bool)
    tool[-1] = True
    recv, send = construct_edges(st, 0.08, mask, tool, topk=5, connect_all=False)
    assert recv.numel() == int(gold["cfg1_n_rel"][0])
    cfg = {str(k): int(v) for k, v in zip(gold["gnn_cfg_keys"], gold["gnn_cfg_vals"])}
    cfg.update(nf_particle=512, nf_relation=512, nf_effect=512)
    torch.manual_seed(0)
    model = DynamicsPredictor(cfg).eval()
    attrs = torch.zeros(1, N, 2)
    attrs[0, :-1, 0] = 1
    attrs[0, -1, 1] = 1
    with torch.no_grad():
        pos, mot = model(state=st[None, None].repeat(1, cfg["n_his"], 1, 1), attrs=attrs, p_instance=torch.ones(1, N - 1, 1),
                         action=torch.tensor(gold["cfg1_action"]), receivers=recv, senders=send)
    np.testing.assert_allclose(pos.numpy(), gold["cfg1_pred_pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mot.numpy(), gold["cfg1_pred_motion"], rtol=2e-4, atol=2e-6)


def test_fit_bone_rotations_vectorised_equals_literal_form():
    """The masked numpy evaluation of the per-bone decision tree against the literal one-bone-at-a-time form, on random chains
    plus the degenerate cases: no neighbours, one neighbour (rank 1), collinear neighbours, planar neighbours (rank 2),
    a reflection (negative determinant)."""
    import numpy as np
    import torch
    from gsdyn.dynamics import _fit_bone_rotations_loop, fit_bone_rotations
    rng = np.random.default_rng(4)
    for trial in range(6):
        nb = 40
        bones = torch.tensor(rng.normal(0, 1, (nb, 3)).astype(np.float32))
        rel = torch.tensor((rng.uniform(0, 1, (nb, nb)) < 0.12).astype(np.int64))
        rel.fill_diagonal_(0)
        rel[0] = 0                                             # no neighbours
        rel[1] = 0; rel[1, 5] = 1                              # one neighbour
        rel[2] = 0; rel[2, 6] = 1; rel[2, 7] = 1
        bones[6] = bones[2] + torch.tensor([0.3, 0.0, 0.0]); bones[7] = bones[2] + torch.tensor([0.7, 0.0, 0.0])   # collinear
        rel[3] = 0; rel[3, 8] = 1; rel[3, 9] = 1; rel[3, 10] = 1
        for k, off in zip((8, 9, 10), ([0.3, 0.1, 0.0], [-0.2, 0.4, 0.0], [0.1, -0.5, 0.0])):                   # planar
            bones[k] = bones[3] + torch.tensor(off)
        ang = 0.3 * (trial + 1)
        Rz = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
        new = bones @ Rz.T + torch.tensor(rng.normal(0, 0.02, (nb, 3)).astype(np.float32))
        if trial % 2:
            new[:, 2] = -new[:, 2]                             # a reflection: negative determinants
        motions = new - bones
        a = fit_bone_rotations(bones, motions, rel)
        b = _fit_bone_rotations_loop(bones, motions, rel)
        assert a.shape == b.shape == (nb, 3, 3)
        assert float((a - b).abs().max()) <= 1e-6, trial
        assert torch.equal(a[0], torch.eye(3))


def test_rollout_matches_reference_loop(golden_dir):
    """``gsdyn.dynamics.rollout`` + ``smooth_frames`` against the reference's ``DynamicsModule.rollout`` and the smoothing block of
    ``collect_scene_data``, captured by importing the reference (tests/golden/gen_rollout_goldens.py): seven steps, two of which
    move the end effector by less than ``dist_thresh`` (repeated frames), history shift, per-step bone re-sampling."""
    from gsdyn.dynamics import DynamicsPredictor, pack_scene_data, rollout, smooth_frames
    z = np.load(os.path.join(golden_dir, "rollout_host.npz"))
    cfg = dict(nf_particle=32, nf_relation=32, nf_effect=32, attr_dim=2, state_dim=0, action_dim=3, pstep=3, rel_attr_dim=2,
               rel_group_dim=1, rel_distance_dim=3, n_his=3)
    torch.manual_seed(0)
    model = DynamicsPredictor(cfg).eval()
    t = lambda k: torch.tensor(z[k])  # noqa: E731
    max_nobj, fps_r, adj, topk, call, dth = z["cfg"]
    S = z["xyz"].shape[0]
    out = rollout(model, t("xyz_0"), t("rgb_0"), t("quat_0"), t("opa_0"), t("eef_xyz"), S, z["inlier"], max_nobj=int(max_nobj),
                  fps_radius_value=float(fps_r), adj_thresh=float(adj), topk=int(topk), connect_all=bool(call), dist_thresh=float(dth))
    xyz, rgb, quat, opa, bones, eef = out
    assert torch.equal(xyz[2], xyz[1]) and torch.equal(xyz[5], xyz[4])          # the two skipped steps repeat their predecessor
    # fp32 rounding of a different operation order (gather / scatter-add message passing, batched rotation fit) compounds over
    # the autoregressive steps: 1e-7 after one step, 7e-6 after five
    np.testing.assert_allclose(xyz.numpy(), z["xyz"], atol=2e-5)
    np.testing.assert_allclose(quat.numpy(), z["quat"], atol=5e-5)
    np.testing.assert_allclose(bones.numpy(), z["bones"], atol=2e-6)
    np.testing.assert_allclose(eef.numpy(), z["eef"], atol=0)
    assert torch.equal(rgb[-1], t("rgb_0")) and torch.equal(opa[-1], t("opa_0"))
    xs, rs, qs, os_, bs, es = smooth_frames(*(a.clone() for a in out))
    np.testing.assert_allclose(xs.numpy(), z["smooth_xyz"], atol=2e-5)
    np.testing.assert_allclose(qs.numpy(), z["smooth_quat"], atol=5e-5)
    np.testing.assert_allclose(bs.numpy(), z["smooth_bones"], atol=2e-6)
    np.testing.assert_allclose(es.numpy(), z["smooth_eef"], atol=1e-7)
    scene, vis = pack_scene_data(xs, rs, qs, os_, torch.full((xs.shape[1], 3), 0.01), bs, es)
    assert len(scene) == S and set(scene[0]) == {"means3D", "colors_precomp", "rotations", "opacities", "scales", "means2D"}
    assert vis[3]["kp"].shape == (int(max_nobj), 3) and vis[3]["tool_kp"].shape == (1, 3)


def test_statistical_outlier_loop_drops_far_points():
    from gsdyn.dynamics import remove_statistical_outliers
    g = torch.Generator().manual_seed(1)
    pts = torch.cat([0.05 * torch.randn(400, 3, generator=g), torch.tensor([[3.0, 0, 0], [0, -4.0, 0], [2.0, 2.0, 2.0]])])
    keep = remove_statistical_outliers(pts, nb_neighbors=20)
    assert keep.max() < 400 and keep.numel() >= 380
