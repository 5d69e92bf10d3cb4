"""Forward-only replica sharding of ``predict.py``'s render loop (SURVEY.md section 8e "Forward-only (config 5)", row E2).

The reference renders, for each of 4 cameras and every frame of an episode, the colour image and a second image with
``colors_precomp = 1`` as an alpha mask (/root/reference/src/predict.py:100-123), one ``Renderer.render`` call each, on one GPU.
Every (frame, camera) pair is independent, so here the pairs are dealt round-robin over the ranks -- pair ``frame * n_cams +
cam`` goes to rank ``(frame * n_cams + cam) mod world`` -- and a rank renders all of its cameras of a frame (colour + mask
each) in ONE multi-view rasterizer call (``Renderer.render_cameras_with_mask``: the two renders of a camera share their tile
lists and are blended in one tile pass).  There is NO collective on the render path; ``gather_frames`` optionally brings
the images to one rank afterwards.  The scene data of a frame (the GNN rollout's output) is the same on every rank: the rollout
is deterministic and cheap next to the renders, so every rank runs it (or rank 0 broadcasts it) -- that part is the caller's.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .camera import look_at_w2c
from .render import Renderer


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
                 bg=(0.0, 0.0, 0.0), near: float = 0.01, far: float = 100.0):
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank, self.world, self.poses, self.bg = int(rank), int(world), list(poses), tuple(bg)
        self.renderer = Renderer(device, w=w, h=h, near=near, far=far)

    def my_pairs(self, n_frames: int) -> List[Tuple[int, int]]:
        return shard_pairs(n_frames, len(self.poses), self.rank, self.world)

    def cams_of_frame(self, frame: int) -> List[int]:
        n = len(self.poses)
        return [c for c in range(n) if (frame * n + c) % self.world == self.rank]

    @torch.no_grad()
    def render_frame(self, frame: int, timestep_data: dict) -> Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """{camera: (image [3,H,W], depth [1,H,W], mask [3,H,W])} for this rank's cameras of ``frame`` -- one rasterizer call."""
        cams = self.cams_of_frame(frame)
        if not cams:
            return {}
        ims, depths, masks = self.renderer.render_cameras_with_mask([self.poses[c] for c in cams], timestep_data, bg=self.bg)
        return {c: (ims[i], depths[i], masks[i]) for i, c in enumerate(cams)}

    @torch.no_grad()
    def render_episode(self, scene_data: Sequence[dict]) -> Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        out = {}
        for f, d in enumerate(scene_data):
            for c, v in self.render_frame(f, d).items():
                out[(f, c)] = v
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
