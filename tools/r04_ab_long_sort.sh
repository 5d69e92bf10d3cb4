cd $GRAFT_REPO_ROOT
for cfg in "GSR_LONG_SORT=2048" "GSR_LONG_SORT=4096"; do
  echo "$cfg: $(env $cfg timeout 300 python bench.py --config 5 --with-rollout --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:round(d[k],4) for k in ('ms_per_step','rollout_ms_per_frame','render_ms_per_frame_this_rank')})")"
  echo "   static: $(env $cfg timeout 300 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['per_kernel_us_per_frame']['tile_sort'])")"
done
