"""Forward-only caller of the path: mirror of the reference's ``Renderer``
(/root/reference/src/render/renderer.py:6-50) -- builds the settings record per call and renders under
``no_grad``; ``predict.py`` drives it per (camera, frame) and renders a second time with colours = 1 as an
alpha mask (/root/reference/src/predict.py:115-123)."""
from __future__ import annotations

import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views

from .camera import setup_camera


class DepthCuts:
    """Speculative per-tile depth cuts for a SEQUENCE of forward-only frames seen by the same cameras (predict.py: four fixed cameras, frame
    after frame of a deforming scene).  In a dense scene most (Gaussian, tile) pairs lie behind the depth at which every pixel of their
    tile has saturated -- 90 % at 500 k Gaussians / 1080p -- and are binned, sorted and never blended.  Every frame's blend proposes, per
    tile, a depth just past what it needed (include/gsr.h: gsr_arm_depth_cuts); the next frame of the same cameras bins only the pairs in
    front of it, and its blend VALIDATES the guess: a cut tile whose list runs out with a pixel still alive is counted in the view's redo
    word.  ``failed()`` returns the (frame, views) to render again without cuts -- a view whose word stayed zero is exact, not
    approximately right.  One object per stream of frames (FrameShard owns one); buffers are kept per camera set.

    ``dilate``: a tile bins with the DEEPEST proposal of its (2 dilate + 1)^2 neighbourhood -- where a silhouette moves by up to 16 dilate
    pixels per frame, the tiles it uncovers inherit "no cut" from a neighbour that already saw the background (default: what the process' last adapting object ended with, 2 at first).  It ADAPTS: the redo words
    of earlier frames come back through pinned memory without a wait (they are read once their copy's event has fired); a frame with a
    failing tile widens the neighbourhood by one tile (up to 6), sixteen clean frames in a row narrow it (down to 1).  Measured on the
    bench episode (Gaussians move 27 pixels per frame: tools/r05_depth_cut_probe.py): dilate 1 fails in every frame (100 tiles of 32 640),
    dilate 4 in 2 of 19 (a handful of tiles), keeping 38 % of the entries instead of 28 %."""
    MAX_TILES = 10240            # GSR_BIN_MAX_T: larger tile grids take the radix binning, which does not cut
    _learned = {"dilate": 2}     # what the last adapting object of this process ended with: predict.py renders episode after episode of one scene
    #                              (/root/reference/src/predict.py:74-164), and the next episode need not find the scene's speed again

    def __init__(self, dilate: int = None, margin: float = 1.01, adapt: bool = True):
        self.dilate, self.margin, self.adapt = int(DepthCuts._learned["dilate"] if dilate is None else dilate), float(margin), bool(adapt)
        self._buf = {}           # camera-set key -> [ping, pong] lists of per-view [T] int32 tensors, index of the last one written (or None)
        self._pending = []       # [frame id, redo words on the device [V], their pinned host copy, the copy's event, seen by the adaptation, dilation used]
        self._pinned = {}        # free pinned slots by word count
        self._clean = 0
        self.calls = self.cut_calls = self.redone = 0

    def _feedback(self):
        for ent in self._pending:
            if not ent[4] and ent[2] is not None and (ent[3] is None or ent[3].query()):      # (no event: host tensors, the words are there)
                ent[4] = True
                if int(ent[2].max()) > 0:
                    if ent[5] >= self.dilate:       # (a failure of a frame armed with a narrower neighbourhood than today's says nothing new:
                        self.dilate = min(self.dilate + 1, 6)    # the words come back a few frames late, and every one of those frames fails)
                    self._clean = 0
                else:
                    self._clean += 1
                    if self._clean >= 16:
                        self.dilate, self._clean = max(self.dilate - 1, 1), 0

    def arm(self, key, n_views: int, h: int, w: int, device, frame_id):
        """(cut_in or None, cut_out, redo, margin) for the next call of camera set ``key``, or None when the image has too many tiles.
        The caller hands ``redo`` to ``sent`` once the call is queued."""
        T = ((h + 15) // 16) * ((w + 15) // 16)
        if T > self.MAX_TILES:
            return None
        if self.adapt:
            self._feedback()
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:       # ("cuda" names the current device: compare like with like below)
            device = torch.device("cuda", torch.cuda.current_device())
        ent = self._buf.get(key)
        if ent is None or ent[0][0][0].numel() != T or ent[0][0][0].device != device or len(ent[0][0]) != n_views:
            if len(self._buf) >= 16:
                self._buf.clear()
            ent = self._buf[key] = [[[torch.empty(T, dtype=torch.int32, device=device) for _ in range(n_views)] for _ in (0, 1)], None]
        bufs, last = ent
        cin = bufs[last] if last is not None else None
        if cin is not None and self.dilate > 0:
            gy, gx = (h + 15) // 16, (w + 15) // 16
            t = torch.stack(cin).view(torch.float32).view(n_views, 1, gy, gx)        # depth bits of positive floats ARE the floats
            t = torch.nn.functional.max_pool2d(t, 2 * self.dilate + 1, stride=1, padding=self.dilate).reshape(n_views, T).view(torch.int32)
            cin = [t[v] for v in range(n_views)]
        nxt = 0 if last is None else 1 - last
        ent[1] = nxt
        redo = torch.zeros(n_views, dtype=torch.int32, device=device)
        self._pending.append([frame_id, redo, None, None, cin is None, self.dilate])   # (a call without cuts cannot fail: nothing to learn from it)
        self.calls += 1
        self.cut_calls += cin is not None
        return cin, bufs[nxt], redo, self.margin

    def sent(self, redo):
        """The call that was armed with ``redo`` is queued: its words travel to pinned memory behind it (no wait here)."""
        for ent in reversed(self._pending):
            if ent[1] is redo:
                if redo.is_cuda:
                    n = int(redo.numel())
                    free = self._pinned.setdefault(n, [])      # pinned slots are recycled (a pinned allocation per call costs ~0.1 ms)
                    ent[2] = free.pop() if free else torch.empty(n, dtype=redo.dtype, pin_memory=True)
                    ent[2].copy_(redo, non_blocking=True)
                    ent[3] = torch.cuda.Event()
                    ent[3].record(torch.cuda.current_stream(redo.device))
                else:
                    ent[2] = redo
                return

    def forget(self, key=None):
        """Drop the hints (of one camera set): the next frame bins everything."""
        if key is None:
            self._buf.clear()
        else:
            self._buf.pop(key, None)

    def failed(self):
        """{frame id: [positions of the call's views whose speculative render must be repeated without cuts]} for the frames since the last
        call (waits for their redo words: they are long there unless the caller asks right behind a render)."""
        bad = {}
        for fid, redo, host, ev, _, _ in self._pending:
            if ev is not None:
                ev.synchronize()
            words = (host if host is not None else redo.cpu()).tolist()
            views = [v for v, x in enumerate(words) if x]
            if views:
                bad[fid] = views
                self.redone += 1
            if host is not None and host is not redo:
                self._pinned.setdefault(int(host.numel()), []).append(host)
        self._pending = []
        if self.adapt:
            DepthCuts._learned["dilate"] = self.dilate
        return bad


class Renderer:
    def __init__(self, device, w: int = 1280, h: int = 720, near: float = 0.01, far: float = 100.0):
        self.near, self.far = near, far
        self.w, self.h = w, h
        self.device = device
        self._cameras = {}

    def _camera(self, w2c, k, bg):
        """The settings record of a camera, built once per distinct (pose, intrinsics, background): ``setup_camera`` costs
        three host->device copies and a device-side 4x4 inverse per call, which the reference pays on every render of every
        frame although predict.py's cameras are fixed (/root/reference/src/render/renderer.py:25-50)."""
        key = (np.asarray(w2c, dtype=np.float64).tobytes(), np.asarray(k, dtype=np.float64).tobytes(), tuple(float(b) for b in bg))
        cam = self._cameras.get(key)
        if cam is None:
            if len(self._cameras) >= 256:
                self._cameras.clear()
            cam = self._cameras[key] = setup_camera(self.w, self.h, k, w2c, near=self.near, far=self.far, bg=bg, device=self.device)
        return cam

    @torch.no_grad()
    def render(self, w2c, k, timestep_data, bg=(0.7, 0.7, 0.7)):
        timestep_data = {key: v.to(self.device) for key, v in timestep_data.items()}
        cam = self._camera(w2c, k, bg)
        im, _, depth = GaussianRasterizer(raster_settings=cam)(**timestep_data)
        return im, depth

    @torch.no_grad()
    def render_with_mask(self, w2c, k, timestep_data, bg=(0.0, 0.0, 0.0)):
        """Colour render + the all-ones 'mask' render of predict.py, as ONE multi-view call: the two renders share the
        camera, so the second one reuses the first one's tile lists (per-view colours, ``geometry_of``)."""
        ims, depths, masks = self.render_cameras_with_mask([(w2c, k)], timestep_data, bg=bg)
        return ims[0], depths[0], masks[0]

    @torch.no_grad()
    def render_cameras_with_mask(self, cameras, timestep_data, bg=(0.0, 0.0, 0.0), mask_from_alpha: bool = True, cuts: "DepthCuts" = None,
                                 cuts_key=None, frame_id=None):
        """All cameras of a frame, colour + mask each, in one rasterizer call (predict.py renders 4 cameras x 2 per
        frame, /root/reference/src/predict.py:100-123).  ``cameras``: list of (w2c, k).  Returns image, depth and mask lists.

        The mask render differs from the colour render in ``colors_precomp = 1`` only (predict.py:119-121), so each of its
        channels is sum_i alpha_i T_i + T_final bg = (1 - T_final) + T_final bg: the colour render's final transmittance says
        it all (sum_i alpha_i T_i telescopes to 1 - T_final over the blended entries, early stop included).  With
        ``mask_from_alpha`` (default) the mask comes from the forward's per-pixel T_final -- ONE plain blend pass per camera
        instead of a six-channel one -- and equals the second render up to fp32 rounding of the two summation orders (~1e-6;
        tests/test_dynamics_gpu.py bounds it against the second render).  ``mask_from_alpha = False`` blends the ones as well
        (fused pair: the two renders share tile lists and records).
        ``cuts`` (with ``mask_from_alpha``): a ``DepthCuts`` object -- the call bins with the previous frame's per-tile depth proposals of camera
        set ``cuts_key`` and registers its redo flags under ``frame_id``; the CALLER asks ``cuts.failed()`` afterwards and renders those frames
        again with ``cuts=None``."""
        d = {key: v.to(self.device) for key, v in timestep_data.items()}
        cams = [self._camera(w2c, k, bg) for w2c, k in cameras]
        n = len(cams)
        col = d["colors_precomp"].float()
        P = d["means3D"].shape[0]
        if mask_from_alpha and P > 0:
            from diff_gaussian_rasterization import _hip
            armed = None
            if cuts is not None and len(set(id(c) for c in cams)) == n:      # (a camera repeated in one call shares its tile lists: not cut)
                armed = cuts.arm(cuts_key if cuts_key is not None else tuple(id(c) for c in cams), n, self.h, self.w, self.device, frame_id)
            out, _, depth, states = _hip.rasterize_forward_batch(
                cams, d["means3D"].float().contiguous(), d["opacities"].float().contiguous(), col.contiguous(), None,
                d["scales"].float().contiguous(), d["rotations"].float().contiguous(), None, prepare_backward=False, forward_only=True,
                depth_cuts=armed)
            if armed is not None:
                cuts.sent(armed[2])
            black = all(float(b) == 0.0 for b in bg)                         # (decided on the host: no device round trip)
            masks = []
            for i in range(n):
                T = _hip.final_transmittance(states[i])                     # [H, W], a view of the state's image buffer
                if black:
                    masks.append((1.0 - T).unsqueeze(0).expand(3, -1, -1))  # three equal channels: no copy
                else:
                    masks.append(1.0 - T.unsqueeze(0) * (1.0 - cams[i].bg.view(3, 1, 1)))
            return [out[i] for i in range(n)], [depth[i] for i in range(n)], masks
        views = [c for c in cams for _ in (0, 1)]
        colours = torch.stack([col, torch.ones_like(col)]).repeat(n, 1, 1)
        out, _, depth = rasterize_gaussians_views(views, d["means3D"], torch.zeros((len(views), P, 3), device=self.device),
                                                  d["opacities"], colors_precomp=colours, scales=d["scales"], rotations=d["rotations"])
        return [out[2 * i] for i in range(n)], [depth[2 * i] for i in range(n)], [out[2 * i + 1] for i in range(n)]
