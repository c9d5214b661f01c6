#!/bin/bash
# Development tool: the LDS-treelet experiment of DESIGN.md section 5 in one GPU call (variants built by
# tools/build_variant.sh treelet85 -DFCG_TREELET=85 and treelet341 -DFCG_TREELET=341): stage times of the shipped library,
# of the shipped library padded to the blocks per CU the treelet leaves (4 and 3), and of the two treelet builds; then SQ / TA
# counter passes of the shipped library and the 85-node treelet.
set -e
L=$GRAFT_REPO_ROOT/ray-tracing_amd/csrc/build/var
echo "== stage ms per 64-spp step: A shipped (5 blocks per CU), B treelet 85 (4 blocks), C treelet 341 (3 blocks)"
timeout -k 10 600 python3 tools/ab_perf.py --reps 2 --spp 64 ray-tracing_amd/libfcgpu.so $L/libfcgpu_treelet85.so $L/libfcgpu_treelet341.so
echo "== shipped library padded to 4 blocks per CU (FCGPU_LDS_PAD_EXTEND = FCGPU_LDS_PAD_SHADOW = 5440)"
FCGPU_LDS_PAD_EXTEND=5440 FCGPU_LDS_PAD_SHADOW=5440 timeout -k 10 300 python3 tools/ab_perf.py --reps 2 --spp 64 ray-tracing_amd/libfcgpu.so
echo "== shipped library padded to 3 blocks per CU (21824)"
FCGPU_LDS_PAD_EXTEND=21824 FCGPU_LDS_PAD_SHADOW=21824 timeout -k 10 300 python3 tools/ab_perf.py --reps 2 --spp 64 ray-tracing_amd/libfcgpu.so
echo "== counters"
bash tools/pmc_variant.sh r03_treelet_pmc treelet85
