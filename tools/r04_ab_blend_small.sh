#!/bin/bash
# round 4: two small blend-kernel experiments on one box -- (a) ticket-length base priority in render_bwd (GSR_BWD_PRIO_LEN), (b) the forward's
# stopped flag in the sign of T (libgsr_fsigned.so, -DGSR_FWD_SIGNED_T)
mkdir -p gpurun_out/r04d; O=gpurun_out/r04d; rm -f $O/ab.txt
( GSR_NO_TORCH_EXT=1 GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/libgsr_fsigned.so python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or random_scenes or early_termination or full_size_matches or fused_pair or per_view or config5_size or forward_only" 2>&1 | tail -3 ) > $O/pytest_fsigned.log 2>&1
( GSR_BWD_PRIO_LEN=300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or random_scenes or bench_step or full_size_prop" 2>&1 | tail -3 ) > $O/pytest_prio.log 2>&1
cat $O/pytest_*.log
for V in 1 2 8; do
  echo "== views $V: render_bwd ticket base priority (entries >= N run at base 1)" >> $O/ab.txt
  bash tools/ab_env.sh "--views $V --no-optimizer" "GSR_BWD_PRIO_LEN=0" "GSR_BWD_PRIO_LEN=300" "GSR_BWD_PRIO_LEN=450" "GSR_BWD_PRIO_LEN=600" >> $O/ab.txt 2>&1
done
for V in 1 4 8; do
  echo "== views $V: forward stopped flag in the sign of T" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_hip.so libgsr_fsigned.so >> $O/ab.txt 2>&1
done
cat $O/ab.txt
