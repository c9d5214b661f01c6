#!/bin/bash
# Development tool: builder / leaf-size A/B on the 255 520-triangle mask stand-in (BASELINE config 4)
set -e
V=$PWD/ray-tracing_amd/csrc/build/var/libfcgpu_leaf4.so
for b in host reference; do
  timeout -k 10 200 python -u tools/config_perf.py mask_scene 2400 3600 8 detail=256 texture_size=1024 bvh=$b | tail -1 | sed "s/^/leaf2 /" | cut -c1-330
  FCGPU_LIBRARY=$V timeout -k 10 200 python -u tools/config_perf.py mask_scene 2400 3600 8 detail=256 texture_size=1024 bvh=$b | tail -1 | sed "s/^/leaf4 /" | cut -c1-330
done
