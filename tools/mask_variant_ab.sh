#!/bin/bash
# Development tool: BASELINE config 4 (mask stand-in, SPILL + TEX kernels) with the default library and a build variant.
# usage: tools/mask_variant_ab.sh VARIANT_NAME
set -e
V=$PWD/ray-tracing_amd/csrc/build/var/libfcgpu_$1.so
for r in 1 2; do
  timeout -k 10 200 python -u tools/config_perf.py mask_scene 3840 2160 8 detail=256 texture_size=1024 | tail -1 | sed "s/^/default /" | cut -c1-330
  FCGPU_LIBRARY=$V timeout -k 10 200 python -u tools/config_perf.py mask_scene 3840 2160 8 detail=256 texture_size=1024 | tail -1 | sed "s/^/$1 /" | cut -c1-330
done
