O=gpurun_out/s2f; mkdir -p $O
for v in 1 0 1 0; do
GSDYN_STEP_FUSED_GLUE=$v python bench.py --config 5 --with-rollout 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$v', {k:round(d[k],3) for k in ('ms_per_step','ms_per_step_overlapped','rollout_ms_per_frame','render_ms_per_frame_this_rank')})"
done | tee $O/fused_ab.txt
