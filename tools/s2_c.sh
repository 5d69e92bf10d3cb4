O=gpurun_out/s2c; mkdir -p $O
( time python -m pytest tests/test_dynamics_gpu.py tests/test_multirank_gpu.py -x -q -m gpu -k "predict or pipelined or packets or episode" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60.json 2> $O/bench_cfg5_ep60.err
tail -4 $O/pytest.log; python -c "
import json
for f in ('bench_cfg5_ep','bench_cfg5_ep60'):
    d=json.load(open('$O/'+f+'.json')); print({k:d[k] for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank','pipeline_parts_ms_per_frame')}); print(d['predicted_ms_per_frame_by_gpus']['8'])
"
