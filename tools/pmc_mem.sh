#!/bin/bash
# Memory-path counters (TA / TCP / TCC / EA) for the bench workload, one or two derived counters per pass.
# History (round 2, gpurun_out/pmc_r02b_mem_base/pass1.log): the first version asked for five derived TCC_EA0_*_sum counters
# in ONE pass; rocprofv3 aborted inside the profiled process during its first copy with
#   rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect
# (a *_sum counter is 16 channel counters; gfx950 cannot collect that many together) and the process then hung until the
# job limit.  Hence: at most two derived counters per pass, every pass under `timeout -k 10 300` IN FRONT of rocprofv3
# (never between rocprofv3 and the program: no exec hop behind the profiler), and the script stops at the first failure.
# usage (on the GPU box, from the repo root): bash tools/pmc_mem.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_WRITE_sum" "TCC_READ_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCC_NORMAL_EVICT_sum TCC_BUSY_sum" "TCC_CYCLE_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-pictures 0 --placement-trials 0 "$@" > $OUT/pass$i.log 2>&1 \
    || { echo "pass $i ($grp) failed or timed out: see $OUT/pass$i.log"; tail -5 $OUT/pass$i.log; exit 1; }
  echo "pass $i done"
done
cd $R
python3 tools/pmc_summary.py $OUT
