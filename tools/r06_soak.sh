#!/bin/bash
# Round 6: every parity-soak stream of round 5 at its round-5 length (seed 77 to 2000 cases), after the blend kernels' conic lost its per-Gaussian
# log2(e) rounding.  Log: gpurun_out/parity_soak.txt (appended by the test); one summary line per stream in gpurun_out/r06_soak/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_soak; mkdir -p $O; rm -f $R/gpurun_out/parity_soak.txt
run() { # cases seed big
  local t0=$(date +%s)
  GSR_SOAK_CASES=$1 GSR_SOAK_SEED=$2 GSR_SOAK_BIG=$3 timeout 3000 python -m pytest $R/tests/test_soak_gpu.py::test_parity_soak -x -q > $O/s$2_b$3.log 2>&1
  echo "cases=$1 seed=$2 big=$3 rc=$? seconds=$(( $(date +%s) - t0 )) :: $(tail -1 $O/s$2_b$3.log)" >> $O/summary.txt
}
run 2000 77 0; run 1400 3 0; run 1000 2025 0; run 1000 4242 0; run 1500 31337 0; run 1500 9 0; run 1500 123 0
run 150 5 1; run 200 6 1
cp $R/gpurun_out/parity_soak.txt $O/parity_soak.txt
cat $O/summary.txt
