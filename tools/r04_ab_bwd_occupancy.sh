#!/bin/bash
# with exact lists the backward builds need <= 82 VGPRs: does a smaller batch at six workgroups per CU pay?  (BB = 80: 25 KB of LDS)
O=gpurun_out/r04o; mkdir -p $O; rm -f $O/ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/gs-dynamics_amd/csrc
for V in 1 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_env.sh "--views $V --no-optimizer" "GSR_X=default" "GSR_BWD_SMALL_BATCH_TILES=0" "GSR_BWD_SMALL_BATCH_TILES=0 GSR_NO_TORCH_EXT=1 GSR_HIP_LIB=$C/libgsr_bb80.so" >> $O/ab.txt 2>&1
done
cut -c1-250 $O/ab.txt | sed 's#/[^ ]*/csrc/##'
