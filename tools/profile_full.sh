#!/bin/bash
# One configuration's complete evidence set on the SHIPPED kernels: default bench line, one traced
# process (its bench line + the profiler's kernel stats describe the same launches), HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) and the SQ instruction / issue / LDS breakdown -- every rocprofv3 pass on its own, under a timeout placed
# IN FRONT of rocprofv3, never combined with a trace domain.
# usage (GPU box, repo root): bash tools/profile_full.sh <tag> [bench args...]
# outputs under gpurun_out/: <tag>_bench.json  <tag>_traced_bench.json  <tag>_kernel_stats.csv  <tag>_pmc_summary.json
#                            <tag>_pmc_sq_breakdown.json
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/pmc_$TAG $OUT/sq_$TAG
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { echo "bench $TAG failed"; tail -5 $OUT/${TAG}_bench.err; exit 1; }
echo "bench $TAG done"
cd /tmp
QUICK="--no-cpu-baseline --e2e-pictures 0 --placement-trials 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py $QUICK "$@" > $OUT/${TAG}_traced_bench.json 2> $OUT/prof_$TAG.log || { echo "trace $TAG failed"; tail -5 $OUT/prof_$TAG.log; exit 1; }
f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
echo "trace $TAG done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$TAG/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 $QUICK "$@" > $OUT/pmc_$TAG/pass$i.log 2>&1 || { echo "pmc pass $i ($grp) failed"; tail -5 $OUT/pmc_$TAG/pass$i.log; exit 1; }
done
(cd $R && python3 tools/pmc_summary.py $OUT/pmc_$TAG > /dev/null && cp $OUT/pmc_$TAG/summary.json $OUT/${TAG}_pmc_summary.json)
echo "pmc $TAG done"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/sq_$TAG/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 $QUICK "$@" > $OUT/sq_$TAG/pass$i.log 2>&1 || { echo "sq pass $i failed"; tail -5 $OUT/sq_$TAG/pass$i.log; exit 1; }
done
(cd $R && python3 tools/pmc_summary.py $OUT/sq_$TAG > /dev/null && cp $OUT/sq_$TAG/summary.json $OUT/${TAG}_pmc_sq_breakdown.json)
echo "sq $TAG done"
