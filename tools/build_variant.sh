#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" [SOURCE]  ->  gs-dynamics_amd/csrc/libgsr_NAME.so  (A/B builds for tools/ab_libs.sh)
# SOURCE = the kernel file rebuilt with the flags (default gsr_render; e.g. gsr_binning); every other object is the product's.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/gs-dynamics_amd/csrc; F=${3:-gsr_render}
make -C $C -j8 >/dev/null
EXTRA=""; { [ "$F" = gsr_preprocess_fwd ] || [ "$F" = gsr_preprocess_bwd ]; } && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $EXTRA $2 -c $C/$F.hip -o /tmp/${F}_$1.o
OBJS=""
for o in gsr_preprocess_fwd gsr_binning gsr_render gsr_preprocess_bwd gsr_loss gsr_dynamics gsr_gnn gsr_rigidity gsr_step gsr_api; do
  if [ $o = $F ]; then OBJS="$OBJS /tmp/${F}_$1.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgsr_$1.so $OBJS
echo built libgsr_$1.so
