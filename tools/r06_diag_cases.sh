#!/bin/bash
# tests/soak_diag.py on a list of "CASE SEED [BIG]" triples (stdin), for each library named in $LIBS -> gpurun_out/r06_diag/<lib>/case_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/gs-dynamics_amd/csrc
cases=$(cat)
for lib in ${LIBS:-libgsr_hip.so}; do
  O=$R/gpurun_out/r06_diag/${lib%.so}; mkdir -p $O
  echo "$cases" | while read -r c s b; do
    [ -z "$c" ] && continue
    GSR_HIP_LIB=$C/$lib GSR_SOAK_BIG=${b:-0} timeout 600 python $R/tests/soak_diag.py $c $s > $O/case_${s}_${c}_b${b:-0}.txt 2>&1
  done
done
