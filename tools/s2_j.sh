O=gpurun_out/s2j; mkdir -p $O
python -m pytest tests/test_depth_cuts_gpu.py tests/test_multiview_gpu.py tests/test_lists_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --config 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60.json 2> $O/bench_cfg5_ep60.err
GSDYN_DEPTH_CUTS=0 python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60_nocuts.json 2> $O/bench_cfg5_ep60_nocuts.err
python -c "
import json
d=json.load(open('$O/bench_cfg5.json')); print({k:round(d[k],3) for k in ('ms_per_step','ms_per_step_depth_cuts')}, d['depth_cuts']['frames_redone'], d['depth_cuts']['per_kernel_us_per_frame']); print(d['roofline']['per_kernel_us_per_frame'])
for f in ('bench_cfg5_ep','bench_cfg5_ep60','bench_cfg5_ep60_nocuts'):
    d=json.load(open('$O/'+f+'.json')); print(f, {k:round(d.get(k),3) for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank')}, d['depth_cuts']['calls_with_cuts'], d['depth_cuts']['frames_redone'])
"
