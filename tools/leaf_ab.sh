#!/bin/bash
# Development tool: leaf-size A/B over the bigger scenes (variant library from tools/build_variant.sh leaf4 -DFCG_MAX_LEAF_PRIMS=4)
set -e
V=$PWD/ray-tracing_amd/csrc/build/var/libfcgpu_leaf4.so
for cfg in "mask_scene 2400 3600 4" "room 1920 1080 8 tess=512" "material_ball_scene 1920 1080 8" "normal_mapped_scene 1920 1080 8"; do
  FCGPU_LIBRARY=$V timeout -k 10 200 python -u tools/config_perf.py $cfg | tail -1 | sed 's/^/leaf4 /'
  timeout -k 10 200 python -u tools/config_perf.py $cfg | tail -1 | sed 's/^/leaf2 /'
done
