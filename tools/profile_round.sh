#!/bin/bash
# Development tool: everything profiles/rNN_*_vK* holds, in one GPU call. usage: tools/profile_round.sh TAG  (e.g. r02_v3)
# Order: PMC passes first (their summary, ray-tracing_amd/pmc_summary.json, is what bench.py prints beside its own timing),
# then the bench runs.
set -e
tag=${1:-r02_vX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh gpurun_out/$tag/pmc 64
python3 tools/pmc_summary.py gpurun_out/$tag/pmc "profiles/${tag}_pmc_{sq,ta_tcp,tcc_a,tcc_b,tcc_c,fetch,write}.txt (tools/pmc_passes.sh: rocprofv3 --pmc over tools/gpu_perf.py 64 64 f32 = 65 spp of the bench workload, batches of 2^27 paths)" 134217728 > $out/pmc_summary.json
cp $out/pmc_summary.json ray-tracing_amd/pmc_summary.json
find $out/pmc -name "*.csv" -delete; find $out/pmc -name "*.db" -delete
timeout -k 10 400 python3 bench.py > $out/bench.json
tail -c 400 $out/bench.json; echo
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 1 > $out/bench_torchrun_n1.json 2> $out/torchrun.err
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 8 --warmup 1 --shard tiles > $out/bench_torchrun_n1_tiles.json 2>> $out/torchrun.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/default_cmd_kernel_stats.csv \;
python3 tools/full_size_launches.py $out/stats > $out/full_size_launches.txt || true
rm -rf $out/stats
head -6 $out/default_cmd_kernel_stats.csv
cat $out/full_size_launches.txt
