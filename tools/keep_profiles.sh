#!/bin/bash
# Development tool: copies what tools/profile_round.sh TAG left under gpurun_out/TAG into profiles/ (tracked) under the
# names profiles/README.md lists. usage: tools/keep_profiles.sh TAG
set -e
tag=$1
src=gpurun_out/$tag
cp $src/bench.json profiles/${tag}_bench.json
cp $src/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp $src/bench_torchrun_n1.json profiles/${tag}_bench_torchrun_n1.json
cp $src/bench_torchrun_n1_tiles.json profiles/${tag}_bench_torchrun_n1_tiles.json
cp $src/default_cmd_kernel_stats.csv profiles/${tag}_bench_default_cmd_kernel_stats.csv
cp $src/full_size_launches.txt profiles/${tag}_bench_full_size_launches.txt
cp $src/pmc_summary.json profiles/${tag}_pmc_summary.json
cp $src/pmc/rays.json profiles/${tag}_pmc_rays.json
for p in sq ta_tcp tcc_a tcc_b tcc_c fetch write; do cp $src/pmc/$p.txt profiles/${tag}_pmc_$p.txt; done
cp $src/pmc_summary.json ray-tracing_amd/pmc_summary.json
ls profiles/${tag}_*
