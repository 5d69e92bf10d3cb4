#!/bin/bash
# Round 5: where does render_bwd's time go?  Ablation builds (results are wrong on purpose; only the kernel times are read):
#   abl_novisit  -- the replay loop walks nothing (staging, barriers, combine, record stores remain)
#   abl_noreduce -- visits keep their arithmetic, the 24-issue wave reduction is replaced by 8 adds
# tools/build_variant.sh abl_novisit -DGSR_ABL_NOVISIT; tools/build_variant.sh abl_noreduce -DGSR_ABL_NOREDUCE
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
{ for v in 4 1; do echo "== views $v"; $R/tools/ab_libs.sh "--views $v --no-optimizer --steps 20 --warmup 5" libgsr_hip.so libgsr_abl_novisit.so libgsr_abl_noreduce.so; done; } > $R/gpurun_out/r05_ablation.txt 2>&1
