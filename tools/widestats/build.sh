#!/bin/bash
# builds tools/widestats/libwidestats.so against the library's host code (run after `make -C ray-tracing_amd/csrc`)
set -e
cd "$(dirname "$0")"
g++ -std=c++17 -O2 -fPIC -shared -o libwidestats.so wide_stats.cpp -L../../ray-tracing_amd -lfcgpu -Wl,-rpath,'$ORIGIN/../../ray-tracing_amd'
echo tools/widestats/libwidestats.so
