O=gpurun_out/s2g; mkdir -p $O
( time python -m pytest tests/test_depth_cuts_gpu.py tests/test_multiview_gpu.py tests/test_dynamics_gpu.py tests/test_multirank_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
python bench.py --config 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
GSDYN_DEPTH_CUTS=0 python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep_nocuts.json 2> $O/bench_cfg5_ep_nocuts.err
python -c "
import json
d=json.load(open('$O/bench_cfg5.json')); print({k:d[k] for k in ('ms_per_step','ms_per_step_depth_cuts','depth_cuts')}); print(d['roofline']['per_kernel_us_per_frame'])
for f in ('bench_cfg5_ep','bench_cfg5_ep_nocuts'):
    d=json.load(open('$O/'+f+'.json')); print(f, {k:d.get(k) for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank')})
"
