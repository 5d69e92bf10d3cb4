"""Diagnostic (GPU box): full-size view 0, HIP vs oracle -- error distribution per gradient tensor and
details of the worst Gaussians."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
from oracle import OracleCamera, TiledOracle

dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
params = synth_scene_params(P, device=dev)
cam = synth_ring_cameras(4, W, H, device=dev)[0]
with torch.no_grad():
    rv0 = params2rendervar(params)
rv = {k: v.detach().clone().requires_grad_(True) for k, v in rv0.items()}
dL = np.random.default_rng(11).uniform(-1, 1, (3, H, W)).astype(np.float32)
im, radii, depth = GaussianRasterizer(raster_settings=cam)(**rv)
im.backward(gradient=torch.tensor(dL, device=dev))
torch.cuda.synchronize()
g = {k: v.detach().cpu().numpy() for k, v in rv.items()}
ocam = OracleCamera(H, W, cam.tanfovx, cam.tanfovy, cam.bg.cpu().numpy(), 1.0, cam.viewmatrix.cpu().numpy().reshape(-1),
                    cam.projmatrix.cpu().numpy().reshape(-1), 0, cam.campos.cpu().numpy())
o2 = TiledOracle(ocam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                 rotations=g["rotations"], nthreads=min(64, os.cpu_count()))
gr = o2.backward(dL)
amb = o2.ambiguous
print("ambiguous px", int(amb.sum()), "colour max abs err (non-amb)", float(np.abs(im.detach().cpu().numpy() - o2.color)[:, ~amb].max()))
print("colour max abs err (all px)", float(np.abs(im.detach().cpu().numpy() - o2.color).max()))
for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
    a = rv[k].grad.detach().cpu().numpy().reshape(P, -1); b = gr[k].reshape(P, -1)
    d = np.abs(a - b).max(1); s = np.abs(b).max()
    idx = np.argsort(-d)[:4]
    q = np.percentile(d / s, [50, 99, 99.9, 99.99])
    print(f"{k:15s} max|b|={s:.3e} rel max={d.max()/s:.3e} p50={q[0]:.1e} p99={q[1]:.1e} p99.9={q[2]:.1e} p99.99={q[3]:.1e} worst idx={idx.tolist()}")
    for i in idx[:2]:
        print(f"      i={i} hip={a[i]} ora={b[i]} opacity={g['opacities'][i,0]:.4f} radius={o2.radii[i]} tiles={o2.tiles_touched[i]} scale={g['scales'][i]}")
