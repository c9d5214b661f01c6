#!/bin/bash
# Development tool: per-launch kernel times (rocprofv3 kernel trace) of builds of libfcgpu.so. usage: per_bounce_ab.sh LIB...
out=$GRAFT_REPO_ROOT/gpurun_out/pb; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  v=$(basename $lib .so)
  export FCGPU_LIBRARY=$GRAFT_REPO_ROOT/$lib
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $out/$v -- python3 tools/gpu_perf.py 64 64 f32 > $out/$v.log 2>&1 || { echo "$v failed"; tail -5 $out/$v.log; exit 1; }
  echo "== $v"; grep "Msamples" $out/$v.log; python3 tools/per_bounce.py $out/$v | grep -E "k_shade<float, 0|k_generate|k_shadow|k_extend"
  rm -rf $out/$v
done
