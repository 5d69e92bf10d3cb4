#!/bin/bash
# Round 6: the soak's recorded misses / tallied rows under the three conic forms of the blend kernels (experiment builds:
# tools/build_variant.sh conicN "-DGSR_CONIC_FORM=N"), then the step's cost, alternating on this box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/gs-dynamics_amd/csrc; O=$R/gpurun_out/r06_conic; mkdir -p $O
for lib in libgsr_conic0.so libgsr_hip.so libgsr_conic2.so; do
  echo "==== $lib" >> $O/diag.txt
  for cs in "671 77" "459 2025" "1255 3" "1373 3" "227 9" "295 123" "98 77" "129 2025" "360 2025"; do
    echo "-- case $cs" >> $O/diag.txt
    GSR_HIP_LIB=$C/$lib timeout 300 python $R/tests/soak_diag.py $cs 2>&1 | grep -E "norm-wise|^case" >> $O/diag.txt
  done
  for cs in "23 6" "156 6"; do
    echo "-- BIG case $cs" >> $O/diag.txt
    GSR_SOAK_BIG=1 GSR_HIP_LIB=$C/$lib timeout 600 python $R/tests/soak_diag.py $cs 2>&1 | grep -E "norm-wise|^case" >> $O/diag.txt
  done
done
bash $R/tools/ab_libs.sh "--config 3" libgsr_conic0.so libgsr_hip.so libgsr_conic2.so > $O/ab.txt 2>&1
tail -12 $O/ab.txt
