#!/bin/bash
# Development tool: the DESIGN.md figures for BASELINE configs 3-5 on one GPU.
set -e
for m in rough_glass smooth_glass diffuse plastic rough_mirror smooth_mirror; do
  timeout -k 10 200 python -u tools/config_perf.py room 1920 1080 16 center=$m | tail -1 | cut -c1-330
done
timeout -k 10 300 python -u tools/config_perf.py mask_scene 2400 3600 8 detail=256 texture_size=1024 | tail -1 | cut -c1-330
timeout -k 10 300 python -u tools/config_perf.py mask_scene 3840 2160 8 detail=256 texture_size=1024 | tail -1 | cut -c1-330
timeout -k 10 300 python -u tools/config_perf.py room 7680 4320 4 | tail -1 | cut -c1-330
