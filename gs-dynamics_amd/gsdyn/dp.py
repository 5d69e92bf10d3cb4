"""View-sharded data parallelism for the tracking step (SURVEY.md section 8e -- new design, the reference
is single-GPU: no torch.distributed anywhere under /root/reference/src).

One process per GPU.  Gaussian parameters are replicated; the views of a step are sharded over ranks
(rank r renders views r, r+world, ...).  The only exchange per optimiser step is ONE all-reduce(SUM) of
a flat fp32 bucket (``GradBucket``): the direct step's backward writes its gradients straight into the bucket's slices
(``render_step_views(grad_out=bucket.views())``: no copy; gradients that come out of autograd as tensors of their own are packed with
one ``torch.cat``), and after the reduce every ``.grad`` is its slice of the bucket; plus one small bucket of densification statistics
(/root/reference/src/tracking/external.py:138-142, /root/reference/src/tracking/train_utils.py:243-245).
On ROCm the ``nccl`` backend is RCCL; the 6.8 MB bucket at 100k Gaussians is latency-bound on the
fully-connected xGMI mesh, so a single call per step is the right shape (no per-tensor collectives).
Every rank then applies the identical Adam update, so replicas stay bit-identical without a broadcast.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .step import LossWeights, get_loss, get_loss_views, loss_and_grads_views


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Round-robin view -> rank assignment: views rank, rank+world, ..."""
    return list(range(rank, num_views, world))


class GradBucket:
    """Flat fp32 buffer for the gradients of all trainable parameters (dict order): ONE all-reduce per step.

    ``zero()`` drops the ``.grad`` tensors (PyTorch's ``zero_grad(set_to_none=True)``), so the first backward of a step hands its
    gradient tensors over without an accumulate kernel and nothing has to be cleared.
    ``views()`` are the bucket's slices shaped like their parameters: a producer that can write its result anywhere -- the rasterizer's
    backward (``gsdyn.step.render_step_views(..., grad_out=bucket.views())``) -- writes the gradients STRAIGHT into the bucket and sets
    ``.grad`` to those slices; ``all_reduce()`` then finds every gradient in place and reduces without a copy.  Gradients that
    arrive as tensors of their own (autograd, ``loss_and_grads_views``) are packed with one ``torch.cat`` as before; parameters without
    a gradient this step contribute zeros (their slice is cleared only when something wrote to the bucket since it was last known
    to be zero).  With a single rank nothing is packed or reduced at all.  (Pre-attached bucket views that autograd ACCUMULATES into,
    the usual DDP layout, cost a fill plus one accumulate kernel per parameter per step -- ~55 us of a 1.2 ms four-view step.)"""

    def __init__(self, params: Dict[str, torch.nn.Parameter]):
        self.names = [k for k, p in params.items() if p.requires_grad]
        self.params = [params[k] for k in self.names]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slices = {}
        off = 0
        for k, p in zip(self.names, self.params):
            n = p.numel()
            self.slices[k] = (off, off + n)
            off += n
        self._views = {k: self.flat[s:e].view_as(p) for (k, (s, e)), p in zip(self.slices.items(), self.params)}
        self._clean = {k: self.flat._version for k in self.names}   # slice k held zeros when the bucket had this version

    def views(self) -> Dict[str, torch.Tensor]:
        """name -> the parameter's slice of the flat buffer (shaped like the parameter): where an in-place producer puts its gradient."""
        return self._views

    def zero(self):
        for p in self.params:
            p.grad = None

    def _in_place(self, k, p) -> bool:
        g = p.grad
        return g is not None and g.data_ptr() == self._views[k].data_ptr() and g.numel() == p.numel() and g.dtype == torch.float32

    def pack(self):
        """Bring every gradient into the flat buffer (missing ones as zeros) and re-point ``.grad`` at it.  Gradients that already ARE
        their bucket slice cost nothing."""
        placed = [self._in_place(k, p) for k, p in zip(self.names, self.params)]
        if self.params and not any(placed):           # nothing was produced in place: one cat over all pieces, as before
            pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in self.params]
            torch.cat(pieces, out=self.flat)
            missing = [k for k, p in zip(self.names, self.params) if p.grad is None]
            self._clean = {k: self.flat._version for k in missing}
        else:
            ver = self.flat._version
            todo = []                                 # (start, end) of slices to clear, adjacent ones merged
            now_clean = []
            for k, p, here in zip(self.names, self.params, placed):
                if here:
                    self._clean.pop(k, None)
                elif p.grad is not None:              # a tensor of its own next to in-place ones
                    self._views[k].copy_(p.grad.reshape(self._views[k].shape))
                    self._clean.pop(k, None)
                else:
                    now_clean.append(k)
                    if self._clean.get(k) != ver:
                        s, e = self.slices[k]
                        if todo and todo[-1][1] == s:
                            todo[-1] = (todo[-1][0], e)
                        else:
                            todo.append((s, e))
            # a producer writing through raw pointers does not bump the version: slices it owns are never in _clean (popped above)
            for s, e in todo:
                self.flat[s:e].zero_()
            for k in now_clean:                       # after every write of this call (copy_ / zero_ bump the version)
                self._clean[k] = self.flat._version
        for k, p in zip(self.names, self.params):
            p.grad = self._views[k]
        return self.flat

    def all_reduce(self, group=None, async_op: bool = False):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self.pack()
            clean = list(self._clean)
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            for k in clean:                      # zeros on every rank sum to zeros: still clean at the buffer's new version
                self._clean[k] = self.flat._version
            return work
        return None


class ViewShardedStep:
    """One optimiser step over a set of views, sharded over the ranks of ``group``."""

    def __init__(self, params, optimizer: Optional[torch.optim.Optimizer], weights: LossWeights = LossWeights(),
                 group=None, batched: Optional[bool] = None, density_control: Optional[dict] = None,
                 density_stats: Optional[bool] = None):
        """``density_control``: dict(remove_thresh, remove_thresh_5k, scale_scene_radius) switches on the reference's
        adaptive density control for first-timestep calls that pass ``iteration`` (``variables`` then needs
        ``scene_radius``); every rank runs it on the all-reduced statistics with the same random seed, so replicas
        stay identical."""
        self.params, self.optimizer, self.weights, self.group = params, optimizer, weights, group
        self.density_control = density_control
        self.density_stats = density_stats     # None: keep the densification statistics in the first timestep only (their only reader)
        if batched is None:   # the multi-view entry point exists in the HIP package (not in the CPU test double)
            import diff_gaussian_rasterization as dgr
            batched = hasattr(dgr, "rasterize_gaussians_views")
        self.batched = batched
        self.bucket = GradBucket(params)
        # colour groups with lr 0 (the reference's tracking schedule): their gradient is never applied, so it is not computed
        lrs = {g.get("name"): g["lr"] for g in optimizer.param_groups} if optimizer is not None else {}
        self.frozen_colours = lrs.get("rgb_colors", 1.0) == 0.0 and lrs.get("seg_colors", 1.0) == 0.0
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, views: Sequence[dict], variables: dict, is_initial_timestep: bool = True,
                 local_only: bool = False, iteration: Optional[int] = None):
        """``views``: all views of the step (every rank passes the same list) unless ``local_only``, in
        which case ``views`` is already this rank's shard.  Returns (sum of local losses, variables)."""
        mine = list(views) if local_only else [views[i] for i in shard_views(len(views), self.rank, self.world)]
        P = self.params["means3D"].shape[0]
        dev = self.params["means3D"].device
        self.bucket.zero()
        # Densification statistics (screen-space gradient norms, seen counts, radii): read only by the density control, i.e.
        # in the first timestep (/root/reference/src/tracking/train_gs.py:31-37) -- the ~15 small kernels and the two extra
        # collectives they cost are skipped at t > 0.
        track = bool(is_initial_timestep) if self.density_stats is None else bool(self.density_stats)
        untouched = None if track else (variables.get("max_2D_radius"), variables.get("seen"))
        stat = torch.zeros((2, P), dtype=torch.float32, device=dev) if track else None   # [grad-norm * seen, seen]
        rad = torch.zeros((P,), dtype=torch.float32, device=dev) if track else None
        total = torch.zeros((), dtype=torch.float32, device=dev)
        if self.batched and mine:
            # all cameras of the shard, colour + segmentation renders, in ONE rasterizer call
            direct = self.frozen_colours and dev.type == "cuda" and 2 * len(mine) <= 16 and \
                (is_initial_timestep or "rev_ptr" in variables)
            if direct:   # the same kernels called back to back, no autograd graph (host time / 3)
                loss, variables, aux = loss_and_grads_views(self.params, mine, variables, is_initial_timestep, self.weights,
                                                            grad_out=self.bucket.views() if self.world > 1 else None)
            else:
                loss, variables, aux = get_loss_views(self.params, mine, variables, is_initial_timestep, self.weights,
                                                      frozen_colours=self.frozen_colours)
                loss.backward()
            total += loss.detach()
            if track:
                with torch.no_grad():
                    seen_v = aux["radii"] > 0                                    # [V,P]
                    g2 = aux["means2D_grad"] if direct else aux["means2D"].grad
                    if g2 is not None:
                        stat[0] += (torch.norm(g2[0::2, :, :2], dim=-1) * seen_v).sum(0)
                    stat[1] += seen_v.sum(0)
                    rad = torch.maximum(rad, variables["max_2D_radius"])
            mine = []
        for data in mine:
            loss, variables = get_loss(self.params, data, variables, is_initial_timestep, self.weights)
            loss.backward()
            total += loss.detach()
            if track:
                with torch.no_grad():
                    seen = variables["seen"]
                    g2 = variables["means2D"].grad
                    if g2 is not None:
                        stat[0] += torch.norm(g2[:, :2], dim=-1) * seen
                    stat[1] += seen
                    rad = torch.maximum(rad, variables["max_2D_radius"])
        self.bucket.all_reduce(self.group)
        if track:
            if self.world > 1:
                dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(rad, op=dist.ReduceOp.MAX, group=self.group)
            with torch.no_grad():
                if "means2D_gradient_accum" in variables:
                    variables["means2D_gradient_accum"] += stat[0]
                    variables["denom"] += stat[1]
                variables["max_2D_radius"] = rad
                variables["seen"] = stat[1] > 0
        else:   # the per-rank values get_loss left behind would differ between replicas: put the shared ones back
            for k, v in zip(("max_2D_radius", "seen"), untouched):
                if v is not None:
                    variables[k] = v
                else:
                    variables.pop(k, None)
        if self.density_control is not None and is_initial_timestep and iteration is not None and self.optimizer is not None:
            # between backward and the optimiser step, as in /root/reference/src/tracking/train_gs.py:31-37
            from .densify import densify
            torch.manual_seed(0x5EED + int(iteration))       # same split offsets on every rank
            before = [id(p) for p in self.bucket.params]
            dc = self.density_control
            densify(self.params, variables, self.optimizer, int(iteration), dc["remove_thresh"], dc["remove_thresh_5k"],
                    dc["scale_scene_radius"], accumulate=False)
            if [id(self.params[k]) for k in self.bucket.names] != before:
                self.bucket = GradBucket(self.params)        # parameters were replaced (cloned / split / pruned / reset)
        if self.optimizer is not None:
            self.optimizer.step()
        return total, variables


def init_variables(P: int, device) -> dict:
    z = lambda: torch.zeros(P, dtype=torch.float32, device=device)  # noqa: E731
    return {"max_2D_radius": z(), "means2D_gradient_accum": z(), "denom": z()}
