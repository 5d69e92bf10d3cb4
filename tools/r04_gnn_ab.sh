#!/bin/bash
# one-launch propagation: coherent accesses vs cache-maintenance fences, and the size of the persistent grid
O=$PWD/gpurun_out/r04s; mkdir -p $O
( timeout 600 python -m pytest tests/test_dynamics_gpu.py -m gpu -x -q -s -k "one_launch or whole_step" 2>&1 | tail -6 ) > $O/pytest.log 2>&1; cat $O/pytest.log
C=$PWD/gs-dynamics_amd/csrc
for cfg in "GSDYN_GNN_FUSED=0" "GSR_GNN_WORKGROUPS=32" "GSR_GNN_WORKGROUPS=64" "GSR_GNN_WORKGROUPS=128" "GSR_GNN_WORKGROUPS=256" "GSR_GNN_WORKGROUPS=128 GSR_NO_TORCH_EXT=1 GSR_HIP_LIB=$C/libgsr_gnnfence.so" "GSR_GNN_WORKGROUPS=32 GSR_NO_TORCH_EXT=1 GSR_HIP_LIB=$C/libgsr_gnnfence.so"; do
  echo "$cfg: $(env $cfg timeout 300 python tools/rollout_graph_loop.py 2>&1 | tail -1)" | sed "s#$C/##"
done 2>&1 | tee $O/ab.txt
