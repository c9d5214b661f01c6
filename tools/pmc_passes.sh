#!/bin/bash
# Development tool: the PMC passes behind profiles/rNN_pmc_*.txt, one rocprofv3 run per counter set
# (at most 8 SQ / 4 TCC / 4 TCP / 2 TA / 2 GRBM counters per run). usage: tools/pmc_passes.sh OUTDIR [SPP]
# The profiled program renders 1 + SPP samples per pixel of the bench workload (default 64: one full-size 2^27-path batch).
set -e
out=${1:-gpurun_out/pmc}
spp=${2:-64}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 170 rocprofv3 --output-format csv --pmc "$@" -d $out/$name -- python3 tools/gpu_perf.py 64 $spp f32 > $out/$name.log 2>&1; python tools/prof_summary.py $out/$name > $out/$name.txt; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run ta_tcp TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE GRBM_COUNT
run tcc_a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcc_b TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum
run tcc_c TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
grep -h "RAYS" $out/sq.log | tail -1 | sed 's/^RAYS //' > $out/rays.json
