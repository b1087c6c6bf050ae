#!/bin/bash
# Builds vqvae_amd/libgeo_hip.so and the instrumented variant tools/abl/libgeo_prof.so (-DGEO_MID_PROF: s_memtime stamps per
# phase of the ConvT2 kernels, read by tools/exp_jvp_ablate.py).  Run in the build container: hipcc cross-compiles for gfx950.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/vqvae_amd/csrc
make -s
mkdir -p $ROOT/tools/abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -DGEO_MID_PROF -c jvp.hip -o /tmp/jvp_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/abl/libgeo_prof.so /tmp/jvp_prof.o build/common.o build/graph.o build/knn.o build/kpp.o build/medoid.o build/prior.o build/sssp.o
ls -la $ROOT/vqvae_amd/libgeo_hip.so $ROOT/tools/abl/libgeo_prof.so
