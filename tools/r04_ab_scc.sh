#!/bin/bash
# tracking bit on the SCC of the blend mask's xor (1 SALU per entry) vs s_cmp + s_addc (2)
O=gpurun_out/r04u; mkdir -p $O; rm -f $O/ab.txt
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or random_scenes or randomised or early_termination or fused_pair or per_view or bench_step" 2>&1 | tail -2 ) > $O/pytest.log 2>&1; cat $O/pytest.log
for V in 1 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_scc0.so libgsr_hip.so >> $O/ab.txt 2>&1
done
cut -c1-220 $O/ab.txt
