O=gpurun_out/s2e; mkdir -p $O; rm -f gpurun_out/parity_soak.txt; R=$PWD
for r in 1 2 3; do for lib in libgsr_hip.so libgsr_oldfps.so; do GSR_HIP_LIB=$R/gs-dynamics_amd/csrc/$lib python tools/fps_thin_timing.py 2>&1 | tail -1; GSR_HIP_LIB=$R/gs-dynamics_amd/csrc/$lib python tools/rollout_graph_loop.py 2>&1 | tail -1; done; done | tee $O/fps_thin_ab.txt
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60.json 2> $O/bench_cfg5_ep60.err
python -c "
import json
for f in ('bench_cfg5_ep','bench_cfg5_ep60'):
    d=json.load(open('$O/'+f+'.json')); print({k:d[k] for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank','pipeline_parts_ms_per_frame')}); print(d['predicted_ms_per_frame_by_gpus']['8'])
"
GSR_SOAK_CASES=600 timeout 2500 python -m pytest tests/test_soak_gpu.py -x -q -m gpu > $O/soak_pytest.log 2>&1; tail -3 $O/soak_pytest.log; tail -1 gpurun_out/parity_soak.txt; grep -c referee gpurun_out/parity_soak.txt
