#!/bin/bash
# tools/build_variant_all.sh NAME "<extra hipcc flags>"  ->  gs-dynamics_amd/csrc/libgsr_NAME.so  (A/B builds for tools/ab_libs.sh)
# Rebuilds the files that see cross-file constants (binning, render, api) with the flags; the rest come from the normal build.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/gs-dynamics_amd/csrc; O=/tmp/gsr_variant_$1; mkdir -p $O
make -C $C -j8 >/dev/null
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $2"
$B -c $C/gsr_binning.hip -o $O/gsr_binning.o &
$B -fno-slp-vectorize -c $C/gsr_render.hip -o $O/gsr_render.o &
$B -c $C/gsr_api.hip -o $O/gsr_api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgsr_$1.so $C/gsr_preprocess_fwd.o $O/gsr_binning.o $O/gsr_render.o $C/gsr_preprocess_bwd.o $C/gsr_loss.o $C/gsr_dynamics.o $C/gsr_gnn.o $C/gsr_rigidity.o $C/gsr_step.o $O/gsr_api.o
echo built libgsr_$1.so
