"""Launch-parameter sweep of the blend / sort kernels -- run on the GPU box:

    python tools/autotune.py                     # the whole grid -> gpurun_out/r05_autotune.json (copied to profiles/ when judged)
    python tools/autotune.py --worker P W H V    # one cell with the current environment: prints one JSON line

Cells: V in {1, 2, 4, 8} views x four densities (the demo's 9 k Gaussians at 640x480, the reference's 1280x720 with 50 k, BASELINE's 100 k at 800^2,
configs[4]'s 500 k at 1920x1080); the step timed is gsdyn.step.render_step_views (fused activations, one multi-view forward, one multi-view backward, all
gradients), per-kernel times from the library's own HIP events.  Candidates: for render_bwd the batch size x workgroups per CU builds
(128 @ 3 / 4 / 5, 96 @ 5, 80 @ 6, 64 @ 7) and the producer / consumer form; for render_fwd the workgroups per CU; for tile_sort the
build the launcher would not pick by itself.  Every candidate is one subprocess (the launchers read their switches once); candidates of a
cell run interleaved three times, the table keeps the minimum of each kernel's time.  The launch heuristics in gsr_render.hip / gsr_binning.hip
cite this table."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gs-dynamics_amd", "csrc")


def worker(P, W, H, V, steps=12):
    import numpy as np
    import torch
    for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
        sys.path.insert(0, p)
    from diff_gaussian_rasterization import _hip
    from gsdyn import synth_ring_cameras, synth_scene_params
    from gsdyn.step import render_step_views
    dev = torch.device("cuda:0")
    params = synth_scene_params(P, seed=0, device=dev)
    cams = synth_ring_cameras(max(V, 4), W, H, device=dev)[:V]
    dL = torch.tensor(np.random.default_rng(1234).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)

    def step():
        render_step_views(params, cams, dL)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    _hip.profile_begin()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    prof = _hip.profile_end()
    print(json.dumps({"step_us_median": ts[len(ts) // 2], "step_us_min": ts[0],
                      "kernels_us": {k: 1e3 * ms / max(n, 1) for k, (ms, n) in prof.items()}}))


CELLS = [(P, W, H, V) for (P, W, H) in ((8_957, 640, 480), (50_000, 1280, 720), (100_000, 800, 800), (500_000, 1920, 1080)) for V in (1, 2, 4, 8)]
BIG = 10 ** 9
ROUNDS = 3          # candidates of a cell run interleaved ROUNDS times; the table keeps each kernel's minimum
# name -> (library build or None, environment).  "default" = what the launchers pick by themselves.
CANDIDATES = {
    "default": (None, {}),
    "bwd128@3": (None, {"GSR_BWD_SMALL_BATCH_TILES": str(BIG), "GSR_BWD_WG_PER_CU": "3"}),
    "bwd128@4": (None, {"GSR_BWD_SMALL_BATCH_TILES": str(BIG), "GSR_BWD_WG_PER_CU": "4"}),
    "bwd80@6": (None, {"GSR_BWD_SMALL_BATCH_TILES": "0"}),
    "bwd96@5": ("libgsr_at_bb96w5.so", {"GSR_BWD_SMALL_BATCH_TILES": "0"}),
    "bwd64@7": ("libgsr_at_bb64w7.so", {"GSR_BWD_SMALL_BATCH_TILES": "0"}),
    "bwd_pc@5": (None, {"GSR_BWD_PC": "1"}),
    "fwd@4": (None, {"GSR_FWD_WG_PER_CU": "4"}),
    "fwd@5": (None, {"GSR_FWD_WG_PER_CU": "5"}),
    "sort1024": (None, {"GSR_TILE_SORT_RCAP": "1024"}),
    "sort2048": (None, {"GSR_TILE_SORT_RCAP": "2048"}),
    "sort4096": (None, {"GSR_TILE_SORT_RCAP": "4096"}),
}
BUILDS = {"libgsr_at_bb96w5.so": "-DBWD_SMALL_BB=96 -DBWD_SMALL_WAVES=5", "libgsr_at_bb64w7.so": "-DBWD_SMALL_BB=64 -DBWD_SMALL_WAVES=7"}


def run_cell(cell, name):
    lib, env_add = CANDIDATES[name]
    env = dict(os.environ, **env_add)
    if lib:
        env["GSR_HIP_LIB"] = os.path.join(CSRC, lib)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"] + [str(x) for x in cell], env=env, timeout=240,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip().splitlines()
        return json.loads(out[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(*[int(x) for x in sys.argv[2:6]])
    cells = CELLS
    if len(sys.argv) > 1 and sys.argv[1] == "--cells":      # e.g. --cells 100000x800x800x4,100000x800x800x1
        cells = [tuple(int(x) for x in c.split("x")) for c in sys.argv[2].split(",")]
    missing = [b for b in BUILDS if not os.path.exists(os.path.join(CSRC, b))]
    if missing:
        raise SystemExit("build the variants first (tools/build_variant.sh at_bb96w5 '%s'; ... at_bb64w7 '%s')" % tuple(BUILDS.values()))
    out_path = os.path.join(ROOT, "gpurun_out", "r05_autotune.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    table, t0 = {}, time.time()
    for cell in cells:
        key = "P%d_%dx%d_V%d" % cell
        res = {}
        for rnd in range(ROUNDS):
            for name in CANDIDATES:
                r = run_cell(cell, name)
                if "error" in r:
                    res.setdefault(name, {})["error"] = r["error"]
                    continue
                cur = res.setdefault(name, {"step_us": r["step_us_median"], "kernels_us": dict(r["kernels_us"])})
                cur["step_us"] = min(cur["step_us"], r["step_us_median"])
                for k, v in r["kernels_us"].items():
                    cur["kernels_us"][k] = min(cur["kernels_us"].get(k, v), v)
        base = res.get("default", {})
        summary = {}
        for kern, prefix in (("render_bwd", "bwd"), ("render_fwd", "fwd"), ("tile_sort", "sort")):
            cands = {n: v["kernels_us"].get(kern) for n, v in res.items() if "kernels_us" in v and (n == "default" or n.startswith(prefix)) and v["kernels_us"].get(kern)}
            if cands:
                best = min(cands, key=cands.get)
                summary[kern] = {"default_us": cands.get("default"), "best": best, "best_us": cands[best], "all_us": cands}
        table[key] = {"cell": dict(zip(("P", "W", "H", "V"), cell)), "default_step_us": base.get("step_us"), "kernels": summary,
                      "step_us_by_candidate": {n: v.get("step_us") for n, v in res.items()}}
        json.dump({"elapsed_s": time.time() - t0, "candidates": {n: {"lib": l, "env": e} for n, (l, e) in CANDIDATES.items()}, "cells": table},
                  open(out_path, "w"), indent=1)
        print(key, {k: (v["best"], round(v["best_us"], 1), round(v["default_us"] or 0, 1)) for k, v in summary.items()}, flush=True)


if __name__ == "__main__":
    main()
