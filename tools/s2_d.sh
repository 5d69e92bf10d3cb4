O=gpurun_out/s2d; mkdir -p $O; rm -f gpurun_out/parity_soak.txt
( time python -m pytest tests/test_dynamics_gpu.py tests/test_multirank_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python tools/rollout_graph_loop.py > $O/graph_loop.txt 2>&1
GSDYN_STEP_FUSED_GLUE=0 python tools/rollout_graph_loop.py > $O/graph_loop_torch_glue.txt 2>&1
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60.json 2> $O/bench_cfg5_ep60.err
tail -4 $O/pytest.log; cat $O/graph_loop.txt $O/graph_loop_torch_glue.txt | grep -v amdgpu
python -c "
import json
for f in ('bench_cfg5_ep','bench_cfg5_ep60'):
    d=json.load(open('$O/'+f+'.json')); print({k:d[k] for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank','pipeline_parts_ms_per_frame')}); print(d['predicted_ms_per_frame_by_gpus']['8'])
"
GSR_SOAK_CASES=600 timeout 2000 python -m pytest tests/test_soak_gpu.py -x -q -m gpu > $O/soak_pytest.log 2>&1; tail -3 $O/soak_pytest.log; tail -1 gpurun_out/parity_soak.txt; grep -c referee gpurun_out/parity_soak.txt
