#!/bin/bash
# Development tool: SQ and TA counter passes of the traversal kernels for the shipped library and for build variants
# (tools/build_variant.sh NAME ...), same workload as tools/pmc_passes.sh. usage: tools/pmc_variant.sh OUTTAG NAME [NAME ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
for v in default "$@"; do
  if [ $v = default ]; then unset FCGPU_LIBRARY; else export FCGPU_LIBRARY=$GRAFT_REPO_ROOT/ray-tracing_amd/csrc/build/var/libfcgpu_$v.so; fi
  timeout -s KILL 170 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq_$v -- python3 tools/gpu_perf.py 64 64 f32 > $out/sq_$v.log 2>&1
  python tools/prof_summary.py $out/sq_$v > $out/sq_$v.txt
  timeout -s KILL 170 rocprofv3 --output-format csv --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $out/ta_$v -- python3 tools/gpu_perf.py 64 64 f32 > $out/ta_$v.log 2>&1
  python tools/prof_summary.py $out/ta_$v > $out/ta_$v.txt
  echo "== $v"
  grep -h "k_extend<float\|k_shadow" $out/sq_$v.txt $out/ta_$v.txt | cut -c1-420
  grep -h "nodes/ray" $out/sq_$v.log || true
done
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
