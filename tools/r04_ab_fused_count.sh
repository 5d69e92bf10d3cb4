#!/bin/bash
# round 4: correctness subset + A/B of the counting form of preprocess_fwd (one box, alternating runs)
mkdir -p gpurun_out/r04b; O=gpurun_out/r04b
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or tile_row or random_scenes or batched or per_view_colours or fused_pair or bench_step or config5 or forward_only or more_views or one_call or direct_step or edge_cases or sees_nothing or last_host" 2>&1 | tail -5 ) > $O/pytest_subset.log 2>&1
cat $O/pytest_subset.log
for V in 1 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_r04base.so libgsr_hip.so libgsr_pc8.so >> $O/ab.txt 2>&1
done
echo "== views 4, env toggle on the new library" >> $O/ab.txt
bash tools/ab_env.sh "--views 4 --no-optimizer" "GSR_NO_FUSED_COUNT=1" "GSR_NO_FUSED_COUNT=0" >> $O/ab.txt 2>&1
for lib in libgsr_r04base.so libgsr_hip.so libgsr_pc8.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg5 ms/frame %.3f' % d['ms_per_step'], d['roofline']['per_kernel_us_per_frame'])" >> $O/ab.txt
done
cat $O/ab.txt
