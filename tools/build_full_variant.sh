#!/bin/bash
# tools/build_full_variant.sh NAME "<extra hipcc flags>"  ->  gs-dynamics_amd/csrc/libgsr_NAME.so with EVERY kernel file rebuilt with the flags
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/gs-dynamics_amd/csrc; D=/tmp/gsr_full_$1; mkdir -p $D
OBJS=""
for o in gsr_preprocess_fwd gsr_binning gsr_render gsr_preprocess_bwd gsr_loss gsr_dynamics gsr_gnn gsr_rigidity gsr_step gsr_api; do
  EXTRA=""; { [ $o = gsr_preprocess_fwd ] || [ $o = gsr_preprocess_bwd ]; } && EXTRA="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $EXTRA $2 -c $C/$o.hip -o $D/$o.o &
  OBJS="$OBJS $D/$o.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgsr_$1.so $OBJS
echo built libgsr_$1.so
