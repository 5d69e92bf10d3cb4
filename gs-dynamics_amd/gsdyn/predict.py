"""Forward-only replica sharding of ``predict.py``'s render loop (SURVEY.md section 8e "Forward-only (config 5)", row E2).

The reference renders, for each of 4 cameras and every frame of an episode, the colour image and a second image with
``colors_precomp = 1`` as an alpha mask (/root/reference/src/predict.py:100-123), one ``Renderer.render`` call each, on one GPU.
Every (frame, camera) pair is independent, so here the pairs are dealt round-robin over the ranks -- pair ``frame * n_cams +
cam`` goes to rank ``(frame * n_cams + cam) mod world`` -- and a rank renders all of its cameras of a frame (colour + mask
each) in ONE multi-view rasterizer call (``Renderer.render_cameras_with_mask``: the two renders of a camera share their tile
lists and are blended in one tile pass).  There is NO collective on the render path; ``gather_frames`` optionally brings
the images to one rank afterwards.  The scene data of a frame (the GNN rollout's output) is needed on every rank, and the rollout is
autoregressive (/root/reference/src/render/dynamics_module.py:53-172: step t + 1 needs step t), so it cannot be sharded by frames.  Two
forms:

* replicated (default; the reference's structure): every rank runs the rollout.  It is NOT cheap next to the renders (round 5: 0.5 - 0.8 ms
  per frame against 1.0 - 1.5 ms of renders on one GPU at 500 k / 1080p), so the episode costs about rollout + render / N per frame -- an
  Amdahl term that caps 8 GPUs at ~2 - 2.7x;
* pipelined (``predict_episode(pipeline=True)``, round 5): ONE rank rolls out and broadcasts what each moving step does to the Gaussians --
  <= ``max_nobj`` bones with a rotation, a translation and a quaternion each: 8.8 KB -- and the other ranks apply it with one skinning
  launch per frame and render (``_predict_episode_pipelined``).  A render rank then costs skinning + render / (N - 1) per frame, the
  producer's rollout is the pipeline's critical path: predicted ~3.2x at 8 GPUs (60 frames) from the parts measured on one
  (``bench.py --config 5 --with-rollout``).  Same frames, bit for bit (tests/test_predict_shard_cpu.py with gloo,
  tests/test_multirank_gpu.py with 2 / 3 processes on the real kernels).

``predict_episode`` composes the whole of /root/reference/src/predict.py:74-164 for one episode -- ``collect_scene_data``
(/root/reference/src/render/dynamics_module.py:174-257: activations, low-opacity and outlier filtering, the autoregressive GNN
``rollout``, the smoothing over repeated frames) followed by this rank's share of the (frame, camera) renders and the RGBA
composition of predict.py:115-128 -- as ONE call per rank.  PNG / ffmpeg output and the keypoint overlay stay with the caller
(out of scope).
"""
from __future__ import annotations

import os

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .camera import look_at_w2c
from .render import DepthCuts, Renderer


def shard_pairs(n_frames: int, n_cams: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """The (frame, camera) pairs of rank ``rank``: pair index frame * n_cams + cam, dealt round-robin."""
    return [(p // n_cams, p % n_cams) for p in range(rank, n_frames * n_cams, world)]


def ring_poses(n_cams: int, w: int, h: int, radius: float = 4.0, height: float = 0.8):
    """(w2c, K) of ``n_cams`` cameras on the benchmark ring (SynthScene-v1: fx = fy = W, principal point at the centre)."""
    k = [[float(w), 0.0, w / 2.0], [0.0, float(w), h / 2.0], [0.0, 0.0, 1.0]]
    return [(look_at_w2c((radius * math.cos(0.3 + 2 * math.pi * i / n_cams), height, radius * math.sin(0.3 + 2 * math.pi * i / n_cams))), k)
            for i in range(n_cams)]


class FrameShard:
    """This rank's share of an episode's renders.  ``poses``: list of (w2c, K) -- predict.py's four cameras."""

    def __init__(self, device, w: int, h: int, poses: Sequence, rank: Optional[int] = None, world: Optional[int] = None,
                 bg=(0.0, 0.0, 0.0), near: float = 0.01, far: float = 100.0, mask_from_alpha: bool = True, speculative: Optional[bool] = None):
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank, self.world, self.poses, self.bg = int(rank), int(world), list(poses), tuple(bg)
        self.renderer = Renderer(device, w=w, h=h, near=near, far=far)
        self.mask_from_alpha = bool(mask_from_alpha)   # the mask render = 1 - final transmittance of the colour render (gsdyn/render.py)
        # Speculative depth cuts (gsdyn.render.DepthCuts; HIP devices): each frame bins only what the previous frame of the same cameras
        # needed, the blend validates it, ``validate`` renders the frames that failed again.  Every frame handed out is exact.
        if speculative is None:
            speculative = torch.device(device).type == "cuda" and os.environ.get("GSDYN_DEPTH_CUTS", "1") != "0"
        self.cuts = DepthCuts() if (speculative and self.mask_from_alpha) else None

    def my_pairs(self, n_frames: int) -> List[Tuple[int, int]]:
        return shard_pairs(n_frames, len(self.poses), self.rank, self.world)

    def cams_of_frame(self, frame: int) -> List[int]:
        n = len(self.poses)
        return [c for c in range(n) if (frame * n + c) % self.world == self.rank]

    @torch.no_grad()
    def render_frame(self, frame: int, timestep_data: dict, exact: bool = False, only=None) -> Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """{camera: (image [3,H,W], depth [1,H,W], mask [3,H,W])} for this rank's cameras of ``frame`` -- one rasterizer call.  With
        speculative depth cuts (``self.cuts``) the result is provisional until ``validate`` has run; ``exact``: no cuts for this call;
        ``only``: positions in this rank's camera list of the frame to render (the views ``validate`` repeats)."""
        cams = self.cams_of_frame(frame)
        if only is not None:
            cams = [cams[i] for i in only]
        if not cams:
            return {}
        ims, depths, masks = self.renderer.render_cameras_with_mask([self.poses[c] for c in cams], timestep_data, bg=self.bg,
                                                                        mask_from_alpha=self.mask_from_alpha,
                                                                        cuts=None if exact else self.cuts, cuts_key=tuple(cams), frame_id=frame)
        return {c: (ims[i], depths[i], masks[i]) for i, c in enumerate(cams)}

    @torch.no_grad()
    def validate(self, frames_out: dict, scene_of_frame, post=None) -> List[int]:
        """After a run of ``render_frame`` calls: the views whose speculative cuts did not hold (``DepthCuts.failed``) are rendered again
        without cuts and replace their entries of ``frames_out`` ({(frame, camera): tensors}); ``scene_of_frame(f)`` -> the frame's render
        inputs, ``post(v)`` -> what the caller stores per pair.  Returns the frames that had such views.  No-op without speculation."""
        if self.cuts is None:
            return []
        bad = self.cuts.failed()
        for f, views in bad.items():
            for c, v in self.render_frame(f, scene_of_frame(f), exact=True, only=views).items():
                frames_out[(f, c)] = v if post is None else post(v)
        return sorted(bad)

    @torch.no_grad()
    def render_episode(self, scene_data: Sequence[dict]) -> Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        out = {}
        for f, d in enumerate(scene_data):
            for c, v in self.render_frame(f, d).items():
                out[(f, c)] = v
        self.validate(out, lambda f: scene_data[f])
        return out


def gather_frames(local: dict, dst: int = 0, group=None) -> Optional[dict]:
    """Optional: all ranks' {(frame, camera): tensors} merged on rank ``dst`` (host copies; not on the render path)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local)
    host = {k: tuple(t.cpu() for t in v) for k, v in local.items()}
    parts = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(host, parts, dst=dst, group=group)
    if parts is None:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


# ------------------------------------------------------------------------------------------ the whole of predict.py for one episode
@torch.no_grad()
def collect_scene_data(model, params: dict, eef_xyz, *, max_nobj: int, fps_radius: float, adj_thresh: float, topk: int, connect_all: bool,
                       dist_thresh: float, n_fps_all: int = 1000, max_steps: int = 1000, low_opacity: float = 0.1,
                       remove_outliers: bool = True, thin_start_idx: int = 0, spatial_sort: bool = True, on_frame=None,
                       on_skin=None, skin_source=None, tracked_only: bool = False, graph_step: bool = True):
    """``DynamicsModule.collect_scene_data`` (/root/reference/src/render/dynamics_module.py:174-257) on the device: ``params`` is
    the tracking result (``params.npz``: means3D [T,P,3] or [P,3], rgb_colors, unnorm_rotations, logit_opacities, log_scales);
    frame 0 is activated, Gaussians with opacity < 0.1 are dropped (:187-192), statistical outliers are excluded from the bone
    sampling (:194-212), then rollout -> smoothing -> per-frame render inputs.  Returns (scene_data, vis_data, timings).
    ``spatial_sort`` (not in the reference): the per-frame arrays are handed to the renderer in Morton order of the frame-0
    positions (``dynamics.spatial_order``: one permutation per episode; the farthest-point picks, which depend on the index order,
    are kept by mapping the inlier list) -- the same images up to exact depth ties, a twice faster entry scatter in the rasterizer.
    ``on_frame(t, frame_dict, event)``: STREAMING mode -- every frame is handed over as soon as it is final (a frame in which the
    Gaussians moved: at once; the repeated frames before it: interpolated right then), with a device event recorded behind its
    last producer on the current stream; the returned scene data are those same per-frame dicts.  Same values as the batch mode
    (the interpolation is ``smooth_segment`` either way).
    ``on_skin`` / ``skin_source``: ``dynamics.rollout``'s two ends of the pipelined episode -- the rank that rolls out hands every moving
    step's skinning packet to ``on_skin``; a rank given a ``skin_source`` runs no outlier filter, no sampling and no network, it only
    moves the Gaussians with the packets it receives (``model`` is not used and may be None).
    ``tracked_only``: the rank that rolls out for OTHERS and neither renders nor returns the scene moves only its tracked particles
    (the ``n_fps_all`` farthest points of the inliers, picked here exactly as the rollout would pick them): sampling, relations, network,
    rotation fit and the packets never look at anything else, and the skinning is per particle -- same packets, same keypoints, bit
    for bit; the returned scene data then hold the tracked particles only."""
    import time
    from . import dynamics as D
    first = lambda t: t[0] if t.dim() == 3 else t   # noqa: E731
    dev = params["means3D"].device
    xyz_0, rgb_0 = first(params["means3D"]).float(), first(params["rgb_colors"]).float()
    quat_0 = torch.nn.functional.normalize(first(params["unnorm_rotations"]).float())
    opa_0 = torch.sigmoid(params["logit_opacities"].float())
    scales_0 = torch.exp(params["log_scales"].float())
    keep = opa_0[:, 0] >= low_opacity
    xyz_0, rgb_0, quat_0, opa_0, scales_0 = xyz_0[keep], rgb_0[keep], quat_0[keep], opa_0[keep], scales_0[keep]
    t0 = time.perf_counter()
    if skin_source is not None:
        inlier = torch.zeros(0, dtype=torch.long, device=dev)          # (the receiving side samples nothing)
    else:
        inlier = D.remove_statistical_outliers(xyz_0) if remove_outliers else torch.arange(xyz_0.shape[0], device=dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    eef = torch.as_tensor(eef_xyz, dtype=torch.float32, device=dev)
    if eef.dim() == 2:
        eef = eef[:, None, :]
    n_steps = min(int(eef.shape[0]), max_steps)
    # Morton order BEFORE the rollout (round 5; after it until then: five 500 k-row gathers per frame, ~0.2 ms of every frame on every
    # rank): the skinning is per Gaussian, and the only index-order-dependent piece -- the farthest-point picks over xyz_0[inlier] --
    # sees the same sequence of points when the inlier list is mapped through the inverse permutation.  Same values, bit for bit, as
    # permuting the finished frames (tests/test_predict_shard_cpu.py).
    if tracked_only and skin_source is None:
        # farthest-point sampling over the picked points, in pick order and from the same start, picks them again in that order
        # (pick k was the farthest of ALL inliers from picks 0 .. k-1, first index on ties: it still is among the picks)
        n_t = min(n_fps_all, int(inlier.shape[0]))
        track = inlier[D.farthest_point_sampler(xyz_0[inlier][None], n_t, start_idx=0)[0]]
        xyz_0, rgb_0, quat_0, opa_0, scales_0 = xyz_0[track], rgb_0[track], quat_0[track], opa_0[track], scales_0[track]
        inlier, spatial_sort = torch.arange(n_t, device=dev), False
    if spatial_sort and int(xyz_0.shape[0]) > 1:
        perm = D.spatial_order(xyz_0)                 # frame 0 IS xyz_0
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        xyz_0, rgb_0, quat_0, opa_0, scales_0 = xyz_0[perm], rgb_0[perm], quat_0[perm], opa_0[perm], scales_0[perm]
        inlier = inv[inlier]
    if on_frame is not None:
        # ---- streaming: frames leave in order as they become final
        scene, state = [], {"cp": 0, "next": 0}

        def emit(arrays, upto):                   # frames state["next"] .. upto are final
            xyz, rgb, quat, opa = arrays[0], arrays[1], arrays[2], arrays[3]
            for t in range(state["next"], upto + 1):
                d = {"means3D": xyz[t], "colors_precomp": rgb[t], "rotations": torch.nn.functional.normalize(quat[t], dim=-1),
                     "opacities": opa[t], "scales": scales_0, "means2D": torch.zeros_like(xyz[t])}
                scene.append(d)
                ev = None
                if dev.type == "cuda":
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                on_frame(t, d, ev)
            state["next"] = upto + 1

        def after_step(i, arrays, repeated):
            if repeated:
                return                              # final only once the next moving frame is known
            D.smooth_segment(arrays, state["cp"], i)
            state["cp"] = i
            emit(arrays, i)

        out = D.rollout(model, xyz_0, rgb_0, quat_0, opa_0, eef, n_steps, inlier, max_nobj=max_nobj, fps_radius_value=fps_radius,
                        adj_thresh=adj_thresh, topk=topk, connect_all=connect_all, dist_thresh=dist_thresh,
                        n_fps_all=min(n_fps_all, int(inlier.shape[0])), thin_start_idx=thin_start_idx, after_step=after_step,
                        on_skin=on_skin, skin_source=skin_source, graph_step=graph_step)
        emit(out, n_steps - 1)                      # trailing repeats of the last moving frame
        kp, tool = out[4].cpu().numpy(), out[5].cpu().numpy()
        vis = [{"kp": kp[t], "tool_kp": tool[t]} for t in range(n_steps)]
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        return scene, vis, {"outlier_filter_ms": (t1 - t0) * 1e3, "rollout_ms": (t2 - t1) * 1e3, "frames": n_steps, "gaussians": int(xyz_0.shape[0])}
    out = D.rollout(model, xyz_0, rgb_0, quat_0, opa_0, eef, n_steps, inlier, max_nobj=max_nobj, fps_radius_value=fps_radius,
                    adj_thresh=adj_thresh, topk=topk, connect_all=connect_all, dist_thresh=dist_thresh,
                    n_fps_all=min(n_fps_all, int(inlier.shape[0])), thin_start_idx=thin_start_idx, on_skin=on_skin, skin_source=skin_source,
                    graph_step=graph_step)
    out = D.smooth_frames(*out)
    scene, vis = D.pack_scene_data(out[0], out[1], out[2], out[3], scales_0, out[4], out[5])
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    return scene, vis, {"outlier_filter_ms": (t1 - t0) * 1e3, "rollout_ms": (t2 - t1) * 1e3, "frames": n_steps, "gaussians": int(xyz_0.shape[0])}


def compose_rgba(im: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """predict.py:125-128 on the device: colour un-premultiplied by the mask render, alpha = mean of the mask's channels.
    [4,H,W], values in [0, 1] scale of the inputs (the reference multiplies by 255 and reverses the channel order for cv2)."""
    return torch.cat([im / (mask + 1e-4), mask.mean(0, keepdim=True)], 0)


@torch.no_grad()
def predict_episode(model, params: dict, eef_xyz, poses: Sequence, w: int, h: int, *, rollout_cfg: dict, rank: Optional[int] = None,
                    world: Optional[int] = None, gather_to: Optional[int] = None, bg=(0.0, 0.0, 0.0), rgba: bool = False,
                    scene_out: Optional[list] = None, overlap: bool = False, pipeline: bool = False, producer: int = 0,
                    producer_renders: bool = False, group=None):
    """One episode of /root/reference/src/predict.py:74-164 on this rank: GNN rollout (every rank, identical), then this rank's
    (frame, camera) pairs -- colour + all-ones mask render per pair, all cameras of a frame in one rasterizer call.
    ``poses``: the cameras as (w2c, K); ``rollout_cfg``: the keyword arguments of ``collect_scene_data`` (max_nobj, fps_radius,
    adj_thresh, topk, connect_all, dist_thresh, ...).  Returns (frames, vis_data, timings): ``frames`` = {(frame, cam): (image,
    depth, mask)} of this rank -- or, with ``gather_to`` = a rank, the merged dict there and None elsewhere; with ``rgba`` the
    image slot holds the composed RGBA instead.  ``scene_out``: a list that receives the per-frame render inputs (the rollout's
    torch ops -- index_add message passing, library GEMMs -- are not bit-reproducible from run to run on a GPU).
    ``overlap`` (HIP devices; opt-in): the renders run on a second stream, issued by a second host thread, WHILE the rollout goes on
    -- the rollout is bound by the host issuing small launches and leaves the GPU idle most of the time, the renders are GPU-bound;
    each frame is rendered as soon as it is final (``collect_scene_data(on_frame=...)``).  Same values as the sequential form.
    Measured at configs[4] size on two MI355X boxes: 2.29 - 2.44 ms per frame on one, 3.8 - 4.0 on the other, against 2.7 - 2.8
    sequential on both -- two host threads that both spin on device synchronisations need the cores for it; hence not the default.
    ``pipeline`` (world > 1): ONE rank (``producer``) rolls out and broadcasts every moving step's skinning packet (``dynamics.pack_skin``:
    22 floats per bone, 8.8 KB at 100 bones); the other ranks only move the Gaussians with the packets (one skinning launch per frame)
    and render -- see ``_predict_episode_pipelined``.  Same frames as the replicated form; the rollout leaves the render ranks' time."""
    import time
    dev = params["means3D"].device
    if pipeline:
        w_ = world if world is not None else (dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1)
        if w_ > 1:
            return _predict_episode_pipelined(model, params, eef_xyz, poses, w, h, rollout_cfg, rank, w_, gather_to, bg, rgba, scene_out,
                                              int(producer), bool(producer_renders), group)
    if overlap and dev.type == "cuda":
        return _predict_episode_overlapped(model, params, eef_xyz, poses, w, h, rollout_cfg, rank, world, gather_to, bg, rgba, scene_out)
    scene, vis, tm = collect_scene_data(model, params, eef_xyz, **rollout_cfg)
    if scene_out is not None:
        scene_out.extend(scene)
    shard = FrameShard(dev, w, h, poses, rank, world, bg=bg)
    t0 = time.perf_counter()
    frames = shard.render_episode(scene)
    if rgba:
        frames = {k: (compose_rgba(v[0], v[2]), v[1], v[2]) for k, v in frames.items()}
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    tm["render_ms"] = (time.perf_counter() - t0) * 1e3
    tm["pairs_on_this_rank"] = len(frames)
    if shard.cuts is not None:
        tm["depth_cut_calls"], tm["frames_redone"] = shard.cuts.cut_calls, shard.cuts.redone
    if gather_to is not None:
        frames = gather_frames(frames, dst=gather_to)
    return frames, vis, tm


def _predict_episode_overlapped(model, params, eef_xyz, poses, w, h, rollout_cfg, rank, world, gather_to, bg, rgba, scene_out):
    """predict_episode with the renders of finished frames overlapping the rollout of later ones: a worker thread with its own HIP
    stream takes (frame, inputs, event) items off a queue, waits for the event on its stream, renders this rank's cameras of the
    frame.  The main thread only rolls out."""
    import queue
    import threading
    import time
    from . import dynamics as _dyn
    dev = params["means3D"].device
    shard = FrameShard(dev, w, h, poses, rank, world, bg=bg)
    todo: "queue.Queue" = queue.Queue()
    frames, failure = {}, []

    def worker():
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(device=dev)
            with torch.no_grad(), torch.cuda.stream(stream):
                while True:
                    item = todo.get()
                    if item is None:
                        break
                    f, d, ev = item
                    if failure:
                        continue
                    if ev is not None:
                        stream.wait_event(ev)
                    for c, v in shard.render_frame(f, d).items():
                        frames[(f, c)] = (compose_rgba(v[0], v[2]), v[1], v[2]) if rgba else v
            stream.synchronize()
        except BaseException as e:      # noqa: BLE001 -- handed to the caller's thread
            failure.append(e)

    th = threading.Thread(target=worker, name="gsdyn-render", daemon=True)
    t0 = time.perf_counter()
    th.start()
    # (a high-priority stream for the rollout's small launches was measured WORSE: 2.81 vs 2.38 ms per frame)
    redo = False
    try:
        scene, vis, tm = collect_scene_data(model, params, eef_xyz, on_frame=lambda f, d, ev: todo.put((f, d, ev)), **rollout_cfg)
    except _dyn.RolloutNeedsHostSVD:
        redo = True
    finally:
        todo.put(None)
        th.join()
    if failure:
        raise failure[0]
    if redo:        # the graphed rollout met a bone only the host's SVD decides after frames had left: drop them, run the eager rollout
        torch.cuda.synchronize(dev)
        return _predict_episode_overlapped(model, params, eef_xyz, poses, w, h, dict(rollout_cfg, graph_step=False), rank, world, gather_to, bg, rgba,
                                           scene_out)
    torch.cuda.synchronize(dev)
    post = (lambda v: (compose_rgba(v[0], v[2]), v[1], v[2])) if rgba else None
    with torch.no_grad():
        shard.validate(frames, lambda f: scene[f], post)            # speculative depth cuts that did not hold: those frames again, exactly
    torch.cuda.synchronize(dev)
    if shard.cuts is not None:
        tm["depth_cut_calls"], tm["frames_redone"] = shard.cuts.cut_calls, shard.cuts.redone
    if scene_out is not None:
        scene_out.extend(scene)
    tm["episode_ms"] = (time.perf_counter() - t0) * 1e3
    tm["render_ms"] = float("nan")                      # not separable: the renders ran under the rollout
    tm["pairs_on_this_rank"] = len(frames)
    tm["overlapped"] = True
    frames = dict(sorted(frames.items()))
    if gather_to is not None:
        frames = gather_frames(frames, dst=gather_to)
    return frames, vis, tm


def render_ranks_of(world: int, producer: int = 0, producer_renders: bool = False) -> List[int]:
    """The ranks of a pipelined episode that render: all of them, or all but the one that rolls out."""
    return [r for r in range(int(world)) if producer_renders or r != int(producer)]


def _predict_episode_pipelined(model, params, eef_xyz, poses, w, h, rollout_cfg, rank, world, gather_to, bg, rgba, scene_out, producer,
                               producer_renders, group):
    """predict_episode with the rollout on ONE rank (the reference has no distributed code: new design).

    The rollout is autoregressive (/root/reference/src/render/dynamics_module.py:99-170: frame t + 1 needs frame t), so it cannot be
    sharded by frames, and replicated on every rank it is an Amdahl term as large as the renders (0.5 - 0.9 ms per frame against 1.0 ms
    at 500 k Gaussians / 1080p x 4 cameras).  But what a step DOES to the Gaussians is tiny: <= ``max_nobj`` bones, each with a rest
    position, a rotation, a translation and a unit quaternion (``dynamics.bone_transforms``) -- the sampling, the relations, the GNN and
    the rotation fit only ever look at the ~1000 tracked particles.  So rank ``producer`` runs ``collect_scene_data`` as before and
    broadcasts each moving step's packet (one ``dist.broadcast`` of 22 x max_nobj + 2 floats; RCCL: the device buffer in stream order,
    no host round trip; gloo: a host copy); every other rank applies the packet to its copy of the previous frame with ONE skinning
    launch (gsr_lbs, written into the frame's slot) -- per Gaussian the same arithmetic on the same inputs, hence the same frames --
    smooths repeated frames exactly as the producer does, and renders its (frame, camera) pairs as soon as a frame is final.  The
    pairs are dealt round-robin over the RENDER ranks (``render_ranks_of``: by default the producer renders nothing -- its rollout is
    the pipeline's critical path).  Per frame a render rank then costs skinning + render / (N - 1) instead of rollout + render / N.
    There is still no collective on the render path; the broadcast is the path's one exchange step."""
    import time
    from . import dynamics as D
    dev = params["means3D"].device
    if rank is None:
        rank = dist.get_rank(group)
    rr = render_ranks_of(world, producer, producer_renders)
    shard = FrameShard(dev, w, h, poses, rr.index(rank), len(rr), bg=bg) if rank in rr else None
    max_nobj = int(rollout_cfg["max_nobj"])
    plen = D.skin_packet_len(max_nobj)
    on_host = dist.get_backend(group) != "nccl"          # gloo (CPU tests, multi-process on one GPU): packets travel as host tensors
    src = producer if group is None else dist.get_global_rank(group, producer)
    frames: Dict[Tuple[int, int], tuple] = {}

    def on_frame(f, d, ev):
        if shard is not None:
            for c, v in shard.render_frame(f, d).items():
                frames[(f, c)] = (compose_rgba(v[0], v[2]), v[1], v[2]) if rgba else v

    # Every rank knows how many packets an attempt carries (packet 0 + one per moving step: the end-effector targets decide that on the
    # host), so a rank that fails on the way can still keep the broadcasts matched: the producer sends empty packets for the steps it did
    # not reach, a render rank drains the ones it did not consume -- and the attempt ends with ONE status word from the producer:
    #   0  the frames stand;
    #   1  the graphed rollout met a bone only the host's SVD decides (dynamics.RolloutNeedsHostSVD) after its packets had left: every rank
    #      drops its frames and the episode runs again with the eager rollout -- what the replicated form does silently on each rank;
    #   2  the producer failed: every rank raises (instead of waiting for a broadcast that never comes).
    eef_t = torch.as_tensor(eef_xyz, dtype=torch.float32)
    eef_t = eef_t[:, None, :] if eef_t.dim() == 2 else eef_t
    n_steps = min(int(eef_t.shape[0]), int(rollout_cfg.get("max_steps", 1000)))
    n_packets = 1 + sum(1 for k in D.moving_steps(eef_t, n_steps, float(rollout_cfg["dist_thresh"]))[1:] if not k)
    pdev = "cpu" if on_host else dev

    def bcast(buf):
        dist.broadcast(buf, src=src, group=group)
        return buf

    t0 = time.perf_counter()
    attempts = 0
    while True:
        count, error, status = [0], None, 0
        graph_step = attempts == 0
        attempts += 1
        if rank == producer:
            def on_skin(i, pk):
                bcast(pk.detach().to("cpu").contiguous() if on_host else pk)     # (RCCL: the current stream waits for the broadcast: the next
                count[0] += 1                                                     #  step's graph replay does not overwrite the packet under it)
            # a producer that renders nothing and hands no scene back only needs its tracked particles (collect_scene_data)
            light = shard is None and scene_out is None
            try:
                scene, vis, tm = collect_scene_data(model, params, eef_xyz, on_frame=None if light else on_frame, on_skin=on_skin,
                                                    tracked_only=light, **dict(rollout_cfg, graph_step=graph_step and rollout_cfg.get("graph_step", True)))
                tm["producer_tracked_only"] = light
            except D.RolloutNeedsHostSVD:
                status = 1
            except BaseException as e:      # noqa: BLE001 -- re-raised below, once the other ranks have been told
                status, error = 2, e
            while count[0] < n_packets:       # (only after a failure: the steps the rollout did not reach)
                on_skin(-1, torch.zeros(plen, dtype=torch.float32, device=pdev))
            bcast(torch.full((1,), float(status), dtype=torch.float32, device=pdev))
        else:
            def skin_source(i):
                count[0] += 1
                return bcast(torch.empty(plen, dtype=torch.float32, device=pdev))
            try:
                scene, vis, tm = collect_scene_data(None, params, eef_xyz, on_frame=on_frame, skin_source=skin_source, **rollout_cfg)
            except BaseException as e:      # noqa: BLE001 -- re-raised below, once this rank's share of the broadcasts has been matched
                error = e
            while count[0] < n_packets:
                skin_source(-1)
            status = int(bcast(torch.empty(1, dtype=torch.float32, device=pdev)).item())
        if error is not None:
            raise error
        if status == 2:
            raise RuntimeError(f"pipelined episode: the rollout on rank {producer} failed (see that rank's traceback)")
        if status == 0:
            break
        if attempts >= 2:
            raise RuntimeError("pipelined episode: the eager rollout asked for a redo (cannot happen: it decides every bone on the host)")
        frames.clear()                        # status 1: these frames came from packets of a rollout that has to be redone
        if shard is not None:
            shard = FrameShard(dev, w, h, poses, rr.index(rank), len(rr), bg=bg)
    tm["rollout_attempts"] = attempts
    if shard is not None:
        post = (lambda v: (compose_rgba(v[0], v[2]), v[1], v[2])) if rgba else None
        with torch.no_grad():
            shard.validate(frames, lambda f: scene[f], post)        # speculative depth cuts that did not hold: those frames again, exactly
        if shard.cuts is not None:
            tm["depth_cut_calls"], tm["frames_redone"] = shard.cuts.cut_calls, shard.cuts.redone
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if scene_out is not None:
        scene_out.extend(scene)
    tm["episode_ms"] = (time.perf_counter() - t0) * 1e3
    tm["render_ms"] = float("nan")                      # not separable: frames are rendered as they become final
    tm["pairs_on_this_rank"] = len(frames)
    tm["pipelined"], tm["producer"], tm["render_ranks"] = True, producer, rr
    frames = dict(sorted(frames.items()))
    if gather_to is not None:
        frames = gather_frames(frames, dst=gather_to, group=group)
    return frames, vis, tm
