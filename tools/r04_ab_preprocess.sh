#!/bin/bash
# round 4: preprocess_fwd variants (LDS-transposed record stores, facing edges, counting form at 2048 / 1024 Gaussians per workgroup); one box
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c; rm -f $O/ab.txt
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or tile_row or random_scenes or batched or bench_step or config5_size or edge_cases or sees_nothing or last_host or torch_extension" 2>&1 | tail -3 ) > $O/pytest_subset.log 2>&1
( GSR_FUSED_COUNT=1 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or tile_row or batched or bench_step or fused_pair or per_view" 2>&1 | tail -3 ) > $O/pytest_subset_fused.log 2>&1
( GSR_FUSED_COUNT=1 GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/libgsr_g1024.so python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or tile_row or batched or bench_step or fused_pair or per_view" 2>&1 | tail -3 ) > $O/pytest_subset_g1024.log 2>&1
cat $O/pytest_subset*.log
for V in 1 4 8; do
  echo "== views $V: base / new (lds stores) / new without lds stores / g1024" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_r04base.so libgsr_hip.so libgsr_nolds.so libgsr_g1024.so >> $O/ab.txt 2>&1
  echo "== views $V: counting form (GSR_FUSED_COUNT=1): 2048 per workgroup / 1024 per workgroup" >> $O/ab.txt
  GSR_FUSED_COUNT=1 bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_hip.so libgsr_g1024.so >> $O/ab.txt 2>&1
done
cat $O/ab.txt
