#!/bin/bash
# Development tool: tools/config_perf.py with the shipped library and a build variant, alternating.
# usage: tools/variant_config_ab.sh VARIANT BUILDER WIDTH HEIGHT SPP [key=value ...]
set -e
v=$1; shift
V=$PWD/ray-tracing_amd/csrc/build/var/libfcgpu_$v.so
for r in 1 2; do
  timeout -k 10 200 python -u tools/config_perf.py "$@" | tail -1 | sed "s/^/shipped /" | cut -c1-330
  FCGPU_LIBRARY=$V timeout -k 10 200 python -u tools/config_perf.py "$@" | tail -1 | sed "s/^/$v /" | cut -c1-330
done
