set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_wide8
mkdir -p $out
for w in 4 8; do
  export FCGPU_BVH_WIDTH=$w
  timeout -s KILL 170 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq$w -- python3 tools/gpu_perf.py 64 64 f32 > $out/sq$w.log 2>&1
  python tools/prof_summary.py $out/sq$w > $out/sq$w.txt
  timeout -s KILL 170 rocprofv3 --output-format csv --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $out/ta$w -- python3 tools/gpu_perf.py 64 64 f32 > $out/ta$w.log 2>&1
  python tools/prof_summary.py $out/ta$w > $out/ta$w.txt
  grep -h "k_extend\|k_shadow" $out/sq$w.txt $out/ta$w.txt | cut -c1-400
  grep -h "RAYS\|nodes/ray" $out/sq$w.log
done
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
