#!/bin/bash
# Memory-path counters (TA / TCP / TCC / EA) for the bench workload.
# CAUTION (round 2): on ROCm 7.2 / gfx950 the first pass of this script made rocprofv3 abort (signal 6 inside the profiled
# process) and the run then sat silent until the job limit -- 8 GPU-minutes for nothing.  Run it under `timeout -k 10 300`
# and one counter group at a time if it is needed again; FETCH_SIZE / WRITE_SIZE / SQ_* (tools/profile_config.sh,
# tools/pmc_profile2.sh) work.
# usage (on the GPU box, from the repo root): bash tools/pmc_mem.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_READ_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-pictures 0 --placement-trials 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py $OUT
