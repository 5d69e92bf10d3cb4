"""TEST DOUBLE: an oracle-backed stand-in for ``diff_gaussian_rasterization._hip`` so that host logic
(autograd glue, get_loss call sequence, data-parallel driver) can be exercised on a GPU-less box.
Lives under tests/ and is installed only by pytest's monkeypatch -- product code never references it."""
import numpy as np
import torch

from oracle import OracleCamera, TiledOracle


class _State:
    pass


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def rasterize_forward(rs, means3D, opacities, colors_precomp, shs, scales, rotations, cov3D_precomp):
    cam = OracleCamera(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                       _np(rs.bg), float(rs.scale_modifier), _np(rs.viewmatrix).reshape(-1),
                       _np(rs.projmatrix).reshape(-1), int(rs.sh_degree), _np(rs.campos), bool(rs.prefiltered))
    o2 = TiledOracle(cam, _np(means3D), _np(opacities), colors_precomp=_np(colors_precomp), shs=_np(shs),
                     scales=_np(scales), rotations=_np(rotations), cov3D_precomp=_np(cov3D_precomp))
    st = _State()
    st.o2, st.H, st.W, st.num_rendered, st.P = o2, cam.image_height, cam.image_width, o2.num_rendered, o2.P
    dev = means3D.device
    return (torch.tensor(o2.color, device=dev), torch.tensor(o2.radii, device=dev),
            torch.tensor(o2.depth, device=dev), st)


def rasterize_backward(state, grad_color, means3D, radii, colors_precomp, shs, scales, rotations, cov3D_precomp,
                       want_color_grad=True):
    g = state.o2.backward(_np(grad_color).astype(np.float32))
    t = lambda a: None if a is None else torch.tensor(a, device=means3D.device)  # noqa: E731
    return (t(g["means3D"]), t(g["means2D"]), t(g["colors_precomp"]) if want_color_grad else None, t(g["opacities"]), t(g["scales"]),
            t(g["rotations"]), t(g["cov3D_precomp"]), t(g["shs"]))


def rasterize_forward_batch(settings_list, means3D, opacities, colors_precomp, shs, scales, rotations, cov3D_precomp,
                            prepare_backward=False, forward_only=False, **_kw):
    """V views, optionally with per-view colours ([V,P,3]) -- one oracle run per view."""
    per_view = colors_precomp is not None and colors_precomp.dim() == 3
    outs = [rasterize_forward(rs, means3D, opacities, colors_precomp[v] if per_view else colors_precomp, shs, scales, rotations,
                              cov3D_precomp) for v, rs in enumerate(settings_list)]
    return (torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.stack([o[2] for o in outs]),
            [o[3] for o in outs])


def final_transmittance(state):
    return torch.tensor(np.asarray(state.o2.final_T))


def rasterize_backward_batch(states, grad_color, means3D, radii, colors_precomp, shs, scales, rotations, cov3D_precomp,
                             want_color_grad=True):
    per_view = colors_precomp is not None and colors_precomp.dim() == 3
    outs = [rasterize_backward(st, grad_color[v], means3D, radii[v], colors_precomp[v] if per_view else colors_precomp, shs,
                               scales, rotations, cov3D_precomp) for v, st in enumerate(states)]
    sm = lambda k: None if outs[0][k] is None else torch.stack([o[k] for o in outs]).sum(0)  # noqa: E731
    d_col = None if outs[0][2] is None else (torch.stack([o[2] for o in outs]) if per_view else sm(2))
    return sm(0), torch.stack([o[1] for o in outs]), d_col, sm(3), sm(4), sm(5), sm(6), sm(7)


def install(monkeypatch):
    from diff_gaussian_rasterization import _hip
    monkeypatch.setattr(_hip, "rasterize_forward", rasterize_forward)
    monkeypatch.setattr(_hip, "rasterize_backward", rasterize_backward)
    monkeypatch.setattr(_hip, "rasterize_forward_batch", rasterize_forward_batch)
    monkeypatch.setattr(_hip, "rasterize_backward_batch", rasterize_backward_batch)
    monkeypatch.setattr(_hip, "final_transmittance", final_transmittance)
