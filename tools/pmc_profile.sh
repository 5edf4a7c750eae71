#!/bin/bash
# Collect rocprofv3 PMC counters for the bench workload, one counter group per pass
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains mixed with --pmc).
# usage (on the GPU box, from the repo root): bash tools/pmc_profile.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-pictures 0 --placement-trials 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py $OUT
