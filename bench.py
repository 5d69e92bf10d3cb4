#!/usr/bin/env python
"""bench.py -- fwd+bwd Mpix/s of the differentiable Gaussian rasterizer on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N>1: launched by torch.distributed.run,
one rank per GPU over RCCL).  One STEP = one pass of the hot path over one batch of synthetic input:
for each of this rank's 4 views of SynthScene-v1 (100 000 Gaussians, 800x800, BASELINE.json configs[2])
``GaussianRasterizer`` forward -> autograd backward with a fixed seeded dL/dcolour (the caller's
parameter activations are applied once, outside the timed region; ``--with-activations`` includes them); for N>1 the step ends with ONE all-reduce of the flat Gaussian-gradient bucket (RCCL).
Weak scaling: every rank renders 4 views (rank r takes cameras 4r..4r+3 of a 4N-camera ring), so
value = 4*N*H*W / t_step.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline":     dominant kernel, algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  "cpu_baseline": the CPU oracle (kind "port": the reference has no CPU path and its CUDA extension is absent)
                  timed on this box's host cores on ONE view fwd+bwd of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

P_GAUSS, W, H, VIEWS_PER_RANK = 100_000, 800, 800, 4   # VIEWS_PER_RANK: --views overrides
HBM_PEAK = 8.0e12       # B/s, MI355X spec (MI355X_MICROARCH.md)
VALU_PEAK = 78.6e12     # fp32 lane-instructions/s (157.3 TFLOP/s / 2)


def algorithmic_bytes(P, D, Npx):
    """SURVEY.md section 8d compulsory traffic per view, per kernel group (bytes)."""
    return {
        "preprocess_fwd": 116 * P, "scan": 8 * P, "emit_entries": 24 * P + 12 * D, "sort": 24 * D,
        "tile_ranges": 8 * D, "render_fwd": 44 * D + 24 * Npx, "render_bwd": 76 * D + 20 * Npx,
        "preprocess_bwd": 140 * P,
        "fwd": 148 * P + 88 * D + 24 * Npx, "bwd": 140 * P + 76 * D + 20 * Npx,
        "total": 288 * P + 164 * D + 44 * Npx,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the get_loss-shaped step and forward-only timings")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE configs[1]-style forward-only timing (extra)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the views of a step are spread over")
    ap.add_argument("--per-view-calls", action="store_true",
                    help="one GaussianRasterizer call per view (reference call pattern) instead of the batched multi-view call")
    ap.add_argument("--with-activations", action="store_true",
                    help="include params2rendervar (normalize/sigmoid/exp) and its backward in the timed step")
    ap.add_argument("--views", type=int, default=4, help="views per rank")
    args = ap.parse_args()
    global VIEWS_PER_RANK
    VIEWS_PER_RANK = args.views

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    single_dev = os.environ.get("GSR_BENCH_SINGLE_DEVICE") == "1"   # smoke test of the N > 1 code path on a 1-GPU box
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:   # all ranks share cuda:0; the collective goes through gloo on a host copy (NOT a perf number)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            _ar = dist.all_reduce

            def _host_all_reduce(t, *a, **k):
                h = t.cpu()
                _ar(h, *a, **k)
                t.copy_(h)
            dist.all_reduce = _host_all_reduce
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    from gsdyn.step import params2rendervar_fused
    from gsdyn.dp import GradBucket

    params = synth_scene_params(P_GAUSS, seed=0, device=dev)
    cams = synth_ring_cameras(VIEWS_PER_RANK * world, W, H, device=dev, first=VIEWS_PER_RANK * rank,
                              count=VIEWS_PER_RANK)
    rng = np.random.default_rng(1234 + rank)
    dLs = [torch.tensor(rng.uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev) for _ in cams]
    bucket = GradBucket(params)
    num_rendered = []

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None

    # The timed step is the rasterizer's own forward + backward.  The caller-side
    # activations (normalize / sigmoid / exp of params2rendervar) are applied ONCE here, outside the timed
    # region, and the rasterizer's input gradients accumulate into these leaves.  --with-activations puts
    # the activations and their autograd backward back inside the step (what train_gs.py executes).
    with torch.no_grad():
        rv_leaf = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    # differentiated inputs of the rasterizer = what the all-reduce bucket carries for N > 1: 14 floats per Gaussian
    # (means2D is a per-view gradient holder for the densification statistics, not a parameter: it stays out)
    for k in ("means3D", "rotations", "opacities", "scales", "colors_precomp"):
        rv_leaf[k].requires_grad_(True)
    leaf_bucket = GradBucket({k: v for k, v in rv_leaf.items() if v.requires_grad})

    def one_view(cam, dL):
        rv = params2rendervar(params) if args.with_activations else rv_leaf
        im, radii, depth = GaussianRasterizer(raster_settings=cam)(**rv)
        if not args.forward_only:
            im.backward(gradient=dL)

    from diff_gaussian_rasterization import rasterize_gaussians_views
    dL_all = torch.stack(dLs)
    m2_views = torch.zeros((len(cams), P_GAUSS, 3), device=dev, requires_grad=True)

    def step(record=False):
        nonlocal streams
        (bucket if args.with_activations else leaf_bucket).zero()
        if not args.per_view_calls:
            rv = params2rendervar_fused(params) if args.with_activations else rv_leaf   # one fused kernel each way (gsr_step.hip)
            im, radii, depth = rasterize_gaussians_views(
                cams, rv["means3D"], m2_views, rv["opacities"], colors_precomp=rv["colors_precomp"], scales=rv["scales"],
                rotations=rv["rotations"])
            if not args.forward_only:
                im.backward(gradient=dL_all)
                m2_views.grad = None
        elif streams is None:
            for cam, dL in zip(cams, dLs):
                one_view(cam, dL)
        else:
            main = torch.cuda.current_stream(dev)
            for i, (cam, dL) in enumerate(zip(cams, dLs)):
                st = streams[i % len(streams)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    one_view(cam, dL)
            for st in streams:
                main.wait_stream(st)
        if world > 1 and not args.forward_only:
            (bucket if args.with_activations else leaf_bucket).all_reduce()

    # capture num_rendered per view once (spy on the backend call; not in the timed region)
    orig = _hip.rasterize_forward

    def spy(*a, **k):
        out = orig(*a, **k)
        num_rendered.append(out[3].num_rendered)
        return out
    orig_b = _hip.rasterize_forward_batch

    def spy_b(*a, **k):
        out = orig_b(*a, **k)
        num_rendered.extend(st.num_rendered for st in out[3])
        return out
    _hip.rasterize_forward, _hip.rasterize_forward_batch = spy, spy_b
    step()
    _hip.rasterize_forward, _hip.rasterize_forward_batch = orig, orig_b
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_step = dt / args.steps
    if world > 1:
        t = torch.tensor([t_step], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_step = float(t.item())

    # ---- per-kernel HIP-event pass (outside the timed region), same call pattern as the timed region: in the
    # batched pattern every stage is one launch for all views of the step on one stream, so nothing overlaps.
    saved = (args.per_view_calls, streams)
    streams = None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _hip.profile_begin()
    prof_steps = max(3, min(10, args.steps))
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = _hip.profile_end()
    args.per_view_calls, streams = saved
    per_launch_us = {k: 1e3 * ms / max(n, 1) for k, (ms, n) in prof.items()}
    per_view_us = {k: 1e3 * ms / (prof_steps * VIEWS_PER_RANK) for k, (ms, n) in prof.items()}
    views_per_launch = 1 if args.per_view_calls else VIEWS_PER_RANK
    busy_us = sum(1e3 * ms for ms, n in prof.values()) / prof_steps

    D = float(np.mean(num_rendered)) if num_rendered else 0.0
    Npx = H * W
    ab = algorithmic_bytes(P_GAUSS, D, Npx)
    mpix = VIEWS_PER_RANK * world * Npx / t_step / 1e6
    path_bytes = VIEWS_PER_RANK * (ab["fwd"] if args.forward_only else ab["total"])
    dom = "render_fwd" if args.forward_only else max(
        (k for k in per_view_us if k in ("render_fwd", "render_bwd")), key=lambda k: per_view_us[k], default="render_bwd")
    dom_us = per_launch_us.get(dom, float("nan"))
    dom_bytes = ab[dom] * views_per_launch      # one launch blends `views_per_launch` views
    dom_achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us == dom_us and dom_us > 0 else None
    pairs = 256.0 * D
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": dom_achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": (dom_achieved / (HBM_PEAK / 1e9)) if dom_achieved else None, "traffic": None,
        "algorithmic_bytes_per_launch": dom_bytes, "views_per_launch": views_per_launch, "avg_launch_us": dom_us,
        "path": {"algorithmic_bytes_per_step_per_gpu": path_bytes, "achieved_GBps": path_bytes / t_step / 1e9,
                 "frac_of_hbm_peak": path_bytes / t_step / HBM_PEAK},
        "valu": {"note": "reference-equivalent rate: 256 pixel-Gaussian pairs per list entry x 25 instruction slots, the "
                         "work the reference's blend loop issues; this path skips most pairs (alpha-box lists + per-quad "
                         "culling), so the figure can exceed the VALU peak",
                 "pixel_gaussian_pairs_per_view": pairs,
                 "render_fwd_reference_equivalent_lane_instr_per_s": (pairs * views_per_launch * 25 / (per_launch_us["render_fwd"] * 1e-6)) if "render_fwd" in per_launch_us else None,
                 "peak_lane_instr_per_s": VALU_PEAK},
        "per_kernel_us_per_view": {k: round(v, 2) for k, v in sorted(per_view_us.items())},
        "per_kernel_us_per_launch": {k: round(v, 2) for k, v in sorted(per_launch_us.items())},
        "gsr_kernels_busy_us_per_step": round(busy_us, 1), "step_us": round(t_step * 1e6, 1),
        "per_kernel_timing": "HIP events around every launch of the library, separate pass after the timed region, same call pattern",
    }

    # measured HBM traffic of the dominant kernel, if a PMC summary of this round is committed (tools/prof_traffic.sh)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if dom in tj:
                roofline["traffic"] = tj[dom]["hbm_bytes_per_launch"] * (views_per_launch / max(1, tj.get("views_per_launch", 1)))
                roofline["traffic_source"] = tj.get("source", "profiles/pmc_traffic.json")
        except Exception:  # noqa: BLE001
            pass

    # VALU issue occupancy of the blend kernels from the committed SQ counters (tools/prof_sq.sh): what actually bounds them
    spath = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(spath):
        try:
            sj = json.load(open(spath))
            occ = {}
            for kname in ("render_fwd", "render_bwd"):
                if kname in sj and kname in per_launch_us and sj.get("views_per_launch", 1) == views_per_launch:
                    slots = per_launch_us[kname] * 1e-6 * 2.4e9 / 4.0 * 1024.0      # 256 CUs x 4 SIMDs, one wave-64 VALU op per 4 cycles
                    occ[kname] = {"valu_wave_instructions": sj[kname].get("SQ_INSTS_VALU"),
                                  "active_valu_quad_cycles": sj[kname].get("SQ_ACTIVE_INST_VALU"),
                                  "issue_slots_in_launch_at_2.4GHz": slots,
                                  "valu_issue_occupancy": sj[kname].get("SQ_ACTIVE_INST_VALU", 0.0) / slots}
            roofline["valu"]["measured"] = occ
            roofline["valu"]["measured_source"] = sj.get("source")
        except Exception:  # noqa: BLE001
            pass

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = run_extras(dev, params, cams, synth_ring_cameras, synth_scene_params)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(params, cams[0], dLs[0], params2rendervar)

    if rank == 0:
        line = {
            "metric": "fwd Mpix/s (forward-only, extra)" if args.forward_only else
                      "fwd+bwd Mpix/s at 100k Gaussians, 4x800^2 views",
            "value": mpix, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: SynthScene-v1, 100k Gaussians, 4 views 800x800 per GPU, "
                                   "colour render fwd+bwd per view" + ("" if world == 1 else ", 1 RCCL all-reduce of the flat grad bucket per step"),
                       "gaussians": P_GAUSS, "views_per_gpu": VIEWS_PER_RANK, "image": [H, W],
                       "num_rendered_per_view": D, "parallelism": f"view-sharded dp{world}",
                       "call_pattern": "per-view GaussianRasterizer calls" if args.per_view_calls else
                                       "one rasterize_gaussians_views call per step (one launch per stage for all views)"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extras": extras,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def _time_ms(fn, iters, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


def run_extras(dev, params, cams, synth_ring_cameras, synth_scene_params):
    """Reported beside the headline: the full get_loss-shaped step of train_gs.py
    (2 renders + SSIM/L1 + rigidity terms + backward, per view) and BASELINE configs[1] (forward only)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import LossWeights, get_loss, get_loss_views, loss_and_grads_views, params2rendervar, synth_targets
    from gsdyn.dp import init_variables
    from gsdyn.step import make_rigidity_variables
    out = {}
    try:
        im_gt, seg_gt = synth_targets(W, H, device=dev)
        variables = init_variables(P_GAUSS, dev)
        variables.update(make_rigidity_variables(params, num_knn=20))
        w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)  # assets/datasets.md weights
        views = [dict(cam=c, im=im_gt, seg=seg_gt, id=i) for i, c in enumerate(cams)]

        def make_step(initial, mode):
            def getloss_step():
                for p in params.values():
                    p.grad = None
                if mode == "all_direct":     # the same library calls back to back, no autograd graph
                    loss_and_grads_views(params, views, variables, initial, w)
                    return
                if mode in ("all", "all_colour_grads"):   # all cameras, colour + seg renders: ONE rasterizer call (8 views)
                    # colour groups have lr 0 in the tracking schedule (train_utils.py:152-164): their gradient is skipped,
                    # which keeps each colour + seg pair one fused tile pass in the backward too
                    loss, _, _ = get_loss_views(params, views, variables, initial, w, frozen_colours=(mode == "all"))
                    loss.backward()
                    return
                for d in views:
                    if mode == "pair_direct":
                        loss_and_grads_views(params, [d], variables, initial, w)
                        continue
                    if mode == "pair":       # the reference's pattern, one camera per iteration: colour + seg as a 2-view call
                        loss, _, _ = get_loss_views(params, [d], variables, initial, w, frozen_colours=True)
                    else:                    # two separate GaussianRasterizer calls per camera, as train_utils.py writes it
                        loss, _ = get_loss(params, d, variables, initial, w)
                    loss.backward()
            return getloss_step
        for name, initial in (("getloss_step_t0", True), ("getloss_step", False)):
            res = {}
            for mode in ("separate", "pair", "pair_direct", "all_colour_grads", "all", "all_direct"):
                ms = _time_ms(make_step(initial, mode), 5, 2)
                res[mode] = {"ms_per_step": ms, "ms_per_view": ms / len(views)}
            out[name] = {"views": len(views), **res["all_direct"], "through_autograd": res["all"],
                         "with_seg_colour_gradient": res["all_colour_grads"],
                         "per_camera_2view_call": res["pair_direct"], "per_camera_2view_call_through_autograd": res["pair"],
                         "separate_calls": res["separate"],
                         "what": "train_gs.py get_loss (colour+seg renders, fused 0.8 L1 + 0.2 (1-SSIM)"
                                 + ("" if initial else ", rigid/rot/iso/floor/bg terms") + ") + backward, "
                                 + ("t = 0" if initial else "t > 0") + "; headline = all cameras in one rasterizer call, library calls back to back (gsdyn.step.loss_and_grads_views)"}
    except Exception as e:  # noqa: BLE001
        out["getloss_step"] = {"error": repr(e)}
    try:   # one whole training iteration of the tracking loop, one camera per iteration as train_gs.py runs it: loss + backward + Adam
        from gsdyn import initialize_optimizer
        res = {}
        for name, initial in (("t0", True), ("t>0", False)):
            for mode in ("reference_shape", "fused"):
                p3 = synth_scene_params(P_GAUSS, seed=0, device=dev)
                v3 = init_variables(P_GAUSS, dev)
                v3.update(make_rigidity_variables(p3, num_knn=20))
                views3 = [dict(cam=c, im=im_gt, seg=seg_gt, id=i) for i, c in enumerate(cams)]
                if mode == "fused":
                    opt = initialize_optimizer(p3, 4.0)                    # gsdyn.optim.FusedAdam on a HIP device
                else:
                    lrs = {g["name"]: g["lr"] for g in initialize_optimizer(p3, 4.0).param_groups}
                    opt = torch.optim.Adam([{"params": [v], "name": k, "lr": lrs[k]} for k, v in p3.items()], lr=0.0, eps=1e-15)
                it = [0]

                def iteration():
                    d = views3[it[0] % len(views3)]
                    it[0] += 1
                    if mode == "fused":
                        loss_and_grads_views(p3, [d], v3, initial, w)
                    else:
                        loss, _ = get_loss(p3, d, v3, initial, w)
                        loss.backward()
                    opt.step()
                    opt.zero_grad(set_to_none=True)
                res[name + " " + mode] = _time_ms(iteration, 20, 5)
        out["train_iteration_one_camera"] = {"ms": res, "what": "train_gs.py iteration (one camera: colour + seg render, losses, backward, Adam step); "
                                             "reference_shape = gsdyn.get_loss (two GaussianRasterizer calls, torch loss ops) + torch.optim.Adam; "
                                             "fused = loss_and_grads_views + gsdyn.optim.FusedAdam"}
    except Exception as e:  # noqa: BLE001
        out["train_iteration_one_camera"] = {"error": repr(e)}
    try:
        p2 = synth_scene_params(50_000, seed=0, device=dev)
        with torch.no_grad():
            rv = {k: v.detach() for k, v in params2rendervar(p2).items()}
        cam = cams[0]

        def fwd():
            with torch.no_grad():
                GaussianRasterizer(raster_settings=cam)(**rv)
        ms = _time_ms(fwd, 20, 5)
        out["forward_only_cfg2"] = {"ms_per_view": ms, "Mpix_per_s": H * W / ms / 1e3,
                                    "what": "BASELINE.json configs[1]: 50k Gaussians, 1 view 800x800, forward only"}
    except Exception as e:  # noqa: BLE001
        out["forward_only_cfg2"] = {"error": repr(e)}
    try:   # predict.py's frame (row A11): every camera rendered twice, colours and an all-ones mask (predict.py:100-123)
        from gsdyn.camera import look_at_w2c
        from gsdyn.render import Renderer
        import math
        rdr = Renderer(dev, w=W, h=H)
        with torch.no_grad():
            data = {k: v.detach() for k, v in params2rendervar(params).items()}      # incl. the means2D holder, as the reference passes it
        kmat = [[float(W), 0.0, W / 2.0], [0.0, float(W), H / 2.0], [0.0, 0.0, 1.0]]
        poses = [(look_at_w2c((4.0 * math.cos(0.3 + 1.57 * i), 0.8, 4.0 * math.sin(0.3 + 1.57 * i))), kmat) for i in range(4)]

        def frame_fused():
            rdr.render_cameras_with_mask(poses, data)

        def frame_reference():
            for w2c, k in poses:
                rdr.render(w2c, k, data, bg=(0.0, 0.0, 0.0))
                ones = dict(data)
                ones["colors_precomp"] = torch.ones_like(data["colors_precomp"])
                rdr.render(w2c, k, ones, bg=(0.0, 0.0, 0.0))
        ms_f, ms_r = _time_ms(frame_fused, 10, 3), _time_ms(frame_reference, 5, 2)
        out["predict_frame_4cams"] = {"ms_per_frame": ms_f, "ms_per_frame_reference_calls": ms_r, "Mpix_per_s": 8 * H * W / ms_f / 1e3,
                                      "what": "predict.py frame: 4 cameras x (colour + all-ones mask render), 100k Gaussians, 800x800, forward only; "
                                              "fused = one multi-view call, each camera's pair blended in one tile pass; reference_calls = 8 Renderer.render calls"}
    except Exception as e:  # noqa: BLE001
        out["predict_frame_4cams"] = {"error": repr(e)}
    try:   # BASELINE.json configs[0]-shaped rollout step on the device (row N4): rope.yaml GNN dims, random weights
        from gsdyn.dynamics import DynamicsPredictor, farthest_point_sampler, rollout_step
        cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
                   rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
        torch.manual_seed(0)
        model = DynamicsPredictor(cfg, device=dev).eval()
        with torch.no_grad():
            rv = {k: v.detach() for k, v in params2rendervar(params).items()}
        t_fps = _time_ms(lambda: farthest_point_sampler(rv["means3D"][None], 1000, start_idx=0), 5, 2)
        pick = farthest_point_sampler(rv["means3D"][None], 100, start_idx=0)[0]
        bones = rv["means3D"][pick]
        hist, eef = bones[None].repeat(3, 1, 1), torch.zeros((3, 1, 3), device=dev)
        t_step = _time_ms(lambda: rollout_step(model, hist, eef, eef[-1] + 0.02, rv["means3D"], rv["rotations"], 0.5, 5), 5, 2)
        out["rollout_step_cfg1"] = {"ms_per_step": t_step, "fps_1000_of_100k_ms": t_fps,
                                    "what": "row N4: relations + DynamicsPredictor (rope.yaml width 512, random weights, 100 bones) + "
                                            "bone fitting (host SVD) + skinning of 100k Gaussians (gsr_lbs); FPS timed separately (gsr_fps)"}
    except Exception as e:  # noqa: BLE001
        out["rollout_step_cfg1"] = {"error": repr(e)}
    return out


def run_cpu_baseline(params, cam, dL, params2rendervar):
    """Oracle O2 (C, OpenMP over tiles) on the host cores: ONE view fwd+bwd of the same workload."""
    try:
        from oracle import OracleCamera, TiledOracle
    except Exception as e:  # noqa: BLE001
        return {"error": f"oracle unavailable: {e}"}
    with torch.no_grad():
        rv = {k: v.detach().cpu().numpy() for k, v in params2rendervar(params).items()}
    ocam = OracleCamera(H, W, cam.tanfovx, cam.tanfovy, cam.bg.cpu().numpy(), 1.0,
                        cam.viewmatrix.cpu().numpy().reshape(-1), cam.projmatrix.cpu().numpy().reshape(-1), 0,
                        cam.campos.cpu().numpy())
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    g = dL.cpu().numpy()
    reps, t0 = 0, time.perf_counter()
    while True:  # bounded sample: whole views until ~20 s of CPU-core time (threads x wall), at most 16 views
        o2 = TiledOracle(ocam, rv["means3D"], rv["opacities"], colors_precomp=rv["colors_precomp"], scales=rv["scales"],
                         rotations=rv["rotations"], nthreads=threads)
        o2.backward(g)
        reps += 1
        dt = time.perf_counter() - t0
        if dt * threads >= 20.0 or reps >= 16:
            break
    return {"value": reps * H * W / dt / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x (1 view 800x800, 100k Gaussians, fwd+bwd) with oracle/gsr_oracle.c, OpenMP over tiles",
            "seconds": dt, "host_cpu_count": cores}


if __name__ == "__main__":
    main()
