#!/bin/bash
# Builds a variant of libfcgpu.so with extra -D flags for same-box A/B timing (tools/ab_perf.py).
# usage: tools/build_variant.sh NAME [-DFLAG=V ...]   ->  ray-tracing_amd/csrc/build/var/libfcgpu_NAME.so
set -e
cd "$(dirname "$0")/../ray-tracing_amd/csrc"
name=$1; shift
make -s all >/dev/null
mkdir -p build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-function -DFCG_ONLY_F32 "$@" -Rpass-analysis=kernel-resource-usage -c -o build/var/fc_kernels_$name.o fc_kernels.hip 2> build/var/resources_$name.txt
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -x c++ -c -o build/var/fc_host_$name.o fc_host.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-function "$@" -x hip -c -o build/var/fc_api_$name.o fc_api.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libfcgpu_$name.so build/var/fc_kernels_$name.o build/fc_kernels_f64.o build/fc_lbvh.o build/var/fc_api_$name.o build/var/fc_host_$name.o
echo build/var/libfcgpu_$name.so
