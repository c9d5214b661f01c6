#!/bin/bash
# Development tool: everything profiles/rNN_*_vK* holds, in one GPU call. usage: tools/profile_round.sh TAG  (e.g. r01_v9)
set -e
tag=${1:-r01_vX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $out/bench.json
tail -c 600 $out/bench.json; echo
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 1 > $out/bench_torchrun_n1.json 2> $out/torchrun.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/default_cmd_kernel_stats.csv \;
head -5 $out/default_cmd_kernel_stats.csv
tools/pmc_passes.sh gpurun_out/$tag/pmc
ls $out/pmc/*.txt
