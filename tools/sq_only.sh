#!/bin/bash
# SQ instruction / issue / LDS counters of the dominant kernel for one bench.py configuration (content ablations included):
# the four SQ passes of tools/profile_full.sh without the traces and traffic passes.
# usage (GPU box, repo root): bash tools/sq_only.sh <tag> [bench args...]   ->  gpurun_out/<tag>_pmc_sq_breakdown.json
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT/sq_$TAG
export TMPDIR=/tmp
cd /tmp
QUICK="--no-cpu-baseline --e2e-pictures 0 --placement-trials 0"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/sq_$TAG/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 $QUICK "$@" > $OUT/sq_$TAG/pass$i.log 2>&1 || { echo "sq pass $i failed"; tail -5 $OUT/sq_$TAG/pass$i.log; [ $i -lt 4 ] && exit 1; }
done
(cd $R && python3 tools/pmc_summary.py $OUT/sq_$TAG > /dev/null && cp $OUT/sq_$TAG/summary.json $OUT/${TAG}_pmc_sq_breakdown.json)
echo "sq $TAG done"
