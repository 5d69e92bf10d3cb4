"""Speculative depth cuts of forward-only frame sequences (ABI 123: gsr_arm_depth_cuts; gsdyn.render.DepthCuts; gsdyn.predict.FrameShard):
a frame bins only the (Gaussian, tile) pairs in front of the per-tile depth the previous frame of the same cameras needed; the blend
validates the guess.  What must hold: a frame whose redo flag stays zero is BIT-IDENTICAL to the uncut render; a frame the cuts do not
serve is flagged, every time; FrameShard hands out exact frames either way."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "gs-dynamics_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

P, W, H, CAMS = 120_000, 640, 368, 4


def _scene(dev, seed=0):
    """A DENSE scene (large, fairly opaque Gaussians: most pixels saturate well before their tile's list ends)."""
    from gsdyn import params2rendervar, synth_scene_params
    params = synth_scene_params(P, seed=seed, device=dev, scale_lo=0.03, scale_hi=0.09)
    with torch.no_grad():
        d = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
        d["opacities"] = d["opacities"].clamp_min(0.6)
    return d


def _entries(states):
    from diff_gaussian_rasterization import _hip
    tot = 0
    for st in states:
        rg = _hip.debug_views(st)["ranges"]
        tot += int((rg[:, 1] - rg[:, 0]).sum())
    return tot


def test_cut_frames_are_bit_identical_or_flagged(dev):
    from diff_gaussian_rasterization import _hip
    from gsdyn.predict import ring_poses
    from gsdyn.render import Renderer
    rdr = Renderer(dev, w=W, h=H)
    d = _scene(dev)
    cams = [rdr._camera(w2c, k, (0.0, 0.0, 0.0)) for w2c, k in ring_poses(CAMS, W, H)]
    args = (cams, d["means3D"].contiguous(), d["opacities"].contiguous(), d["colors_precomp"].contiguous(), None, d["scales"].contiguous(),
            d["rotations"].contiguous(), None)
    want = _hip.rasterize_forward_batch(*args, forward_only=True)
    full = _entries(want[3])
    T = ((H + 15) // 16) * ((W + 15) // 16)
    new = lambda: [torch.full((T,), -1, dtype=torch.int32, device=dev) for _ in range(CAMS)]  # noqa: E731
    z = lambda: torch.zeros(CAMS, dtype=torch.int32, device=dev)                                 # noqa: E731
    INF = 0x7f800000
    # frame 0: no cuts in, proposals out -- the uncut render, and a proposal for every tile
    c0, r0 = new(), z()
    got = _hip.rasterize_forward_batch(*args, forward_only=True, depth_cuts=(None, c0, r0))
    assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and torch.equal(got[1], want[1]) and int(r0.max()) == 0
    assert _entries(got[3]) == full and all(int(c.min()) > 0 for c in c0)
    finite = sum(int((c != INF).sum()) for c in c0)
    assert finite > 0.5 * CAMS * T, (finite, CAMS * T)          # the scene is dense: most tiles finish before their lists do
    # frame 1, same scene, binned with frame 0's proposals: far fewer entries, the same images bit for bit, no flag; its proposals = frame 0's
    c1, r1 = new(), z()
    got = _hip.rasterize_forward_batch(*args, forward_only=True, depth_cuts=(c0, c1, r1))
    kept = _entries(got[3])
    assert int(r1.max()) == 0 and kept < 0.6 * full, (kept, full)
    assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and torch.equal(got[1], want[1])
    for v in range(CAMS):       # final transmittance / contributor counts (the mask image comes from them) as well
        a, b = _hip.debug_views(got[3][v]), _hip.debug_views(want[3][v])
        assert torch.equal(a["final_T"], b["final_T"]) and torch.equal(a["n_contrib"], b["n_contrib"])
    assert all(torch.equal(a, b) for a, b in zip(c0, c1))
    # a scene the cuts do NOT serve: the nearest Gaussians of every camera fade away, so pixels now reach deeper than the cuts allow
    d2 = {k: v.clone() for k, v in d.items()}
    d2["opacities"] = torch.where(torch.rand(P, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) < 0.7,
                                  torch.full_like(d["opacities"], 0.02), d["opacities"])
    args2 = (cams, d2["means3D"].contiguous(), d2["opacities"].contiguous(), d2["colors_precomp"].contiguous(), None, d2["scales"].contiguous(),
             d2["rotations"].contiguous(), None)
    want2 = _hip.rasterize_forward_batch(*args2, forward_only=True)
    c2, r2 = new(), z()
    got2 = _hip.rasterize_forward_batch(*args2, forward_only=True, depth_cuts=(c1, c2, r2))
    assert int(r2.min()) >= 1, r2.tolist()                       # every camera: flagged (the words count the failing tiles)
    assert not torch.equal(got2[0], want2[0])                    # (and rightly so: the cut images ARE wrong here)
    # the proposals of the flagged frame are still sound for the NEXT frame of that scene: exact again, or flagged -- never silently wrong
    c3, r3 = new(), z()
    got3 = _hip.rasterize_forward_batch(*args2, forward_only=True, depth_cuts=(c2, c3, r3))
    for v in range(CAMS):
        assert int(r3[v]) >= 1 or (torch.equal(got3[0][v], want2[0][v]) and torch.equal(got3[2][v], want2[2][v])), v
    assert int(r3.sum()) == 0                                    # (here: the flagged tiles proposed no cut, the others' proposals held)
    # cuts of +inf everywhere = no cuts
    cinf = [torch.full((T,), INF, dtype=torch.int32, device=dev) for _ in range(CAMS)]
    c4, r4 = new(), z()
    got4 = _hip.rasterize_forward_batch(*args, forward_only=True, depth_cuts=(cinf, c4, r4))
    assert _entries(got4[3]) == full and torch.equal(got4[0], want[0]) and int(r4.max()) == 0
    # the arming is one-shot: the next plain call is uncut
    again = _hip.rasterize_forward_batch(*args, forward_only=True)
    assert _entries(again[3]) == full and torch.equal(again[0], want[0])
    # a call that is not forward-only refuses cuts
    with pytest.raises(ValueError):
        _hip.rasterize_forward_batch(*args, forward_only=False, depth_cuts=(None, new(), z()))


def test_frame_shard_hands_out_exact_frames(dev):
    """FrameShard with speculative cuts over a drifting scene with one abrupt change: every (frame, camera) result equals the exact
    render of that frame, the abrupt frame is among the repeated ones, and most frames were served by cuts."""
    from gsdyn.predict import FrameShard, ring_poses
    d = _scene(dev, seed=3)
    frames = []
    g = torch.Generator(device=dev).manual_seed(5)
    fade = torch.rand(P, 1, device=dev, generator=g) < 0.6
    for f in range(8):
        e = {k: v.clone() for k, v in d.items()}
        e["means3D"] = d["means3D"] + 0.002 * f                               # a slow drift ...
        if f >= 5:
            e["opacities"] = torch.where(fade, torch.full_like(d["opacities"], 0.02), d["opacities"])   # ... and an abrupt change at frame 5
        frames.append(e)
    poses = ring_poses(CAMS, W, H)
    exact = FrameShard(dev, W, H, poses, rank=0, world=1, speculative=False).render_episode(frames)
    shard = FrameShard(dev, W, H, poses, rank=0, world=1, speculative=True)
    assert shard.cuts is not None
    got = shard.render_episode(frames)
    assert sorted(got) == sorted(exact)
    for k in exact:
        for i in range(3):
            assert torch.equal(got[k][i], exact[k][i]), (k, i)
    assert shard.cuts.calls == 8 and shard.cuts.cut_calls == 7 and 1 <= shard.cuts.redone <= 4 and shard.cuts.dilate >= 2, (shard.cuts.calls, shard.cuts.cut_calls, shard.cuts.redone, shard.cuts.dilate)
    # two ranks' shares (camera subsets differ per frame parity): still exact
    for r in range(2):
        part = FrameShard(dev, W, H, poses, rank=r, world=2, speculative=True).render_episode(frames)
        for k, v in part.items():
            assert all(torch.equal(v[i], exact[k][i]) for i in range(3)), (r, k)


def test_depth_cuts_policy_edge_cases(dev):
    """DepthCuts / FrameShard off the beaten path: a device named "cuda" (no index), another Gaussian count from one frame to the next, a
    changed camera set under the same key, an image with more tiles than the tile-row binning serves -- always the exact frames."""
    from gsdyn.predict import FrameShard, ring_poses
    from gsdyn.render import DepthCuts, Renderer
    d = _scene(dev, seed=7)
    poses = ring_poses(CAMS, W, H)
    exact = FrameShard(dev, W, H, poses, rank=0, world=1, speculative=False)
    want = exact.render_episode([d, d])
    # "cuda" instead of cuda:0: the proposal buffers must survive from frame to frame (cut_calls counts the calls that binned with cuts)
    shard = FrameShard("cuda", W, H, poses, rank=0, world=1, speculative=True)
    got = shard.render_episode([d, d, d])
    assert shard.cuts.cut_calls == 2
    for k in want:
        assert all(torch.equal(got[k][i], want[k][i]) for i in range(3)), k
    # fewer Gaussians in the next frame (the per-tile proposals do not care), then a different scene altogether
    half = {k: (v[: P // 2].contiguous() if v.shape[0] == P else v) for k, v in d.items()}
    other = _scene(dev, seed=11)
    seq = [d, half, other, other]
    got = FrameShard(dev, W, H, poses, rank=0, world=1, speculative=True).render_episode(seq)
    ref = exact.render_episode(seq)
    for k in ref:
        assert all(torch.equal(got[k][i], ref[k][i]) for i in range(3)), k
    # the same key with other cameras behind it (a caller that keys by position): validated like any bad guess
    rdr = Renderer(dev, w=W, h=H)
    dc = DepthCuts()
    a = rdr.render_cameras_with_mask(poses[:2], d, cuts=dc, cuts_key="k", frame_id=0)
    b = rdr.render_cameras_with_mask(poses[2:], d, cuts=dc, cuts_key="k", frame_id=1)      # binned with the OTHER cameras' proposals
    bad = dc.failed()
    b_exact = rdr.render_cameras_with_mask(poses[2:], d)
    for v in range(2):
        assert v in bad.get(1, []) or all(torch.equal(b[j][v], b_exact[j][v]) for j in range(3)), v
    assert 0 not in bad and len(a[0]) == 2
    # 4K: 240 x 135 tiles > 10 240 -- no cuts are armed, the call is the plain one
    big = Renderer(dev, w=3840, h=2160)
    dc2 = DepthCuts()
    few = {k: (v[:20000].contiguous() if v.shape[0] == P else v) for k, v in d.items()}
    big.render_cameras_with_mask(ring_poses(1, 3840, 2160), few, cuts=dc2, cuts_key="k", frame_id=0)
    assert dc2.calls == 0 and dc2.failed() == {}


@pytest.mark.parametrize("seed", range(10))
def test_random_sequences_exact_or_flagged(dev, seed):
    """Seeded random sequences -- Gaussian count, image size (ragged tile edges included), Gaussian size, opacity floor, camera count and
    the motion from frame to frame (drift, jitter, a rotation about the scene's axis, opacity flicker, mixtures) all vary -- binned with
    the PREVIOUS frame's raw proposals, no dilation: plenty of bad guesses.  The one invariant: a view whose redo word stays zero equals
    the uncut render bit for bit (colour, depth, final transmittance, contributor counts); proposals never name a depth of 0."""
    from diff_gaussian_rasterization import _hip
    from gsdyn import params2rendervar, synth_scene_params
    from gsdyn.predict import ring_poses
    from gsdyn.render import Renderer
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([3000, 20000, 60000]))
    w, h = int(rng.integers(40, 500)), int(rng.integers(40, 400))
    cams_n = int(rng.integers(1, 5))
    lo = float(rng.choice([0.01, 0.03, 0.06]))
    params = synth_scene_params(n, seed=seed, device=dev, scale_lo=lo, scale_hi=3 * lo)
    with torch.no_grad():
        d = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
        d["opacities"] = d["opacities"].clamp_min(float(rng.choice([0.05, 0.4, 0.8])))
    rdr = Renderer(dev, w=w, h=h)
    cams = [rdr._camera(w2c, k, (0.0, 0.0, 0.0)) for w2c, k in ring_poses(cams_n, w, h)]
    T = ((h + 15) // 16) * ((w + 15) // 16)
    g = torch.Generator(device=dev).manual_seed(seed)
    prev, clean, flagged = None, 0, 0
    for f in range(6):
        kind = int(rng.integers(0, 5))
        with torch.no_grad():
            if kind == 0:
                d["means3D"] = d["means3D"] + float(rng.uniform(0.0, 0.01))
            elif kind == 1:
                d["means3D"] = d["means3D"] + float(rng.uniform(0.0, 0.02)) * torch.randn(n, 3, device=dev, generator=g)
            elif kind == 2:
                a = float(rng.uniform(0.0, 0.05))
                rot = torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], device=dev, dtype=torch.float32)
                d["means3D"] = d["means3D"] @ rot.T
            elif kind == 3:
                keep = torch.rand(n, 1, device=dev, generator=g) > float(rng.uniform(0.0, 0.5))
                d["opacities"] = torch.where(keep, d["opacities"], torch.full_like(d["opacities"], 0.02))
            # kind 4: the frame repeats
        args = (cams, d["means3D"].contiguous(), d["opacities"].contiguous(), d["colors_precomp"].contiguous(), None, d["scales"].contiguous(),
                d["rotations"].contiguous(), None)
        want = _hip.rasterize_forward_batch(*args, forward_only=True)
        cout = [torch.full((T,), -1, dtype=torch.int32, device=dev) for _ in range(cams_n)]
        redo = torch.zeros(cams_n, dtype=torch.int32, device=dev)
        got = _hip.rasterize_forward_batch(*args, forward_only=True, depth_cuts=(prev, cout, redo))
        for v in range(cams_n):
            assert int(cout[v].min()) > 0, (f, v)
            if int(redo[v]) == 0:
                a, b = _hip.debug_views(got[3][v]), _hip.debug_views(want[3][v])
                assert torch.equal(got[0][v], want[0][v]) and torch.equal(got[2][v], want[2][v]) and torch.equal(got[1][v], want[1][v]), (f, v, kind)
                assert torch.equal(a["final_T"], b["final_T"]) and torch.equal(a["n_contrib"], b["n_contrib"]), (f, v, kind)
                clean += 1
            else:
                assert f > 0
                flagged += 1
        prev = cout
    print(f"depth-cut sequence {seed}: n={n} {w}x{h} cams={cams_n}: {clean} views exact under cuts, {flagged} flagged")
    assert clean >= cams_n         # (frame 0 at least: it had no cuts)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "6"))))
def test_random_episodes_through_frame_shard(dev, seed):
    """The whole policy layer on seeded random episodes (drift, jitter, rotation, opacity flicker, repeats; one to three ranks' shares):
    FrameShard with speculative cuts hands out exactly the frames of FrameShard without them."""
    from gsdyn import params2rendervar, synth_scene_params
    from gsdyn.predict import FrameShard, ring_poses
    rng = np.random.default_rng(3300 + seed)
    n = int(rng.choice([20000, 80000]))
    w, h, cams_n = int(rng.integers(100, 520)), int(rng.integers(80, 400)), int(rng.integers(1, 5))
    lo = float(rng.choice([0.02, 0.05]))
    params = synth_scene_params(n, seed=seed, device=dev, scale_lo=lo, scale_hi=3 * lo)
    with torch.no_grad():
        d = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
        d["opacities"] = d["opacities"].clamp_min(float(rng.choice([0.3, 0.7])))
    g = torch.Generator(device=dev).manual_seed(seed)
    frames = []
    for f in range(10):
        kind = int(rng.integers(0, 5))
        e = {k: v.clone() for k, v in (frames[-1] if frames else d).items()}
        if kind == 0:
            e["means3D"] = e["means3D"] + float(rng.uniform(0.0, 0.01))
        elif kind == 1:
            e["means3D"] = e["means3D"] + float(rng.uniform(0.0, 0.01)) * torch.randn(n, 3, device=dev, generator=g)
        elif kind == 2:
            a = float(rng.uniform(0.0, 0.04))
            rot = torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], device=dev, dtype=torch.float32)
            e["means3D"] = e["means3D"] @ rot.T
        elif kind == 3:
            keep = torch.rand(n, 1, device=dev, generator=g) > float(rng.uniform(0.0, 0.4))
            e["opacities"] = torch.where(keep, e["opacities"], torch.full_like(e["opacities"], 0.02))
        frames.append(e)
    poses = ring_poses(cams_n, w, h)
    exact = FrameShard(dev, w, h, poses, rank=0, world=1, speculative=False).render_episode(frames)
    world = int(rng.integers(1, 4))
    seen = set()
    for r in range(world):
        shard = FrameShard(dev, w, h, poses, rank=r, world=world, speculative=True)
        got = shard.render_episode(frames)
        for k, v in got.items():
            assert k not in seen and all(torch.equal(v[i], exact[k][i]) for i in range(3)), (seed, r, world, k)
            seen.add(k)
    assert seen == set(exact)
