#!/bin/bash
# Development tool: config_perf over several scenes with and without an environment setting. usage: tools/env_ab.sh "NAME=VALUE"
set -e
for cfg in "room 1920 1080 32" "mask_scene 2400 3600 4" "room 1920 1080 8 tess=512" "material_ball_scene 1920 1080 8" "glass_scene 1920 1080 8"; do
  timeout -k 10 200 python -u tools/config_perf.py $cfg | tail -1 | sed 's/^/base /' | cut -c1-330
  env "$1" timeout -k 10 200 python -u tools/config_perf.py $cfg | tail -1 | sed "s/^/$1 /" | cut -c1-330
done
