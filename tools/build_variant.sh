#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags for gsr_render.hip>"  ->  gs-dynamics_amd/csrc/libgsr_NAME.so  (A/B builds for tools/ab_libs.sh)
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/gs-dynamics_amd/csrc
make -C $C -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $2 -c $C/gsr_render.hip -o /tmp/gsr_render_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libgsr_$1.so $C/gsr_preprocess_fwd.o $C/gsr_binning.o /tmp/gsr_render_$1.o $C/gsr_preprocess_bwd.o $C/gsr_loss.o $C/gsr_dynamics.o $C/gsr_rigidity.o $C/gsr_step.o $C/gsr_api.o
echo built libgsr_$1.so
