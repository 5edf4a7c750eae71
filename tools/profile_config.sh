#!/bin/bash
# One configuration's evidence set: bench JSON line, rocprofv3 --kernel-trace --stats summary, and the HBM traffic
# counters (FETCH_SIZE and WRITE_SIZE in separate --pmc passes; never mixed with trace domains).
# usage (on the GPU box, from the repo root): bash tools/profile_config.sh <tag> [bench args...]
# (the profiled passes run the kernel-path leg only: --e2e-pictures 0 keeps the pipeline's launches out of the summaries)
# outputs: gpurun_out/<tag>_bench.json (the default command), <tag>_traced_bench.json + <tag>_kernel_stats.csv (one traced
# process: its line and the profiler's averages describe the same launches), <tag>_pmc_summary.json
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/pmc_$TAG
export TMPDIR=/tmp
cd $R
python3 bench.py "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench $TAG done"
cd /tmp
# the traced run prints its own bench line: the same launches are behind <tag>_traced_bench.json and <tag>_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --e2e-pictures 0 --placement-trials 0 "$@" > $OUT/${TAG}_traced_bench.json 2> $OUT/prof_$TAG.log
f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
echo "trace $TAG done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$TAG/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-pictures 0 --placement-trials 0 "$@" > $OUT/pmc_$TAG/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_$TAG > /dev/null
cp $OUT/pmc_$TAG/summary.json $OUT/${TAG}_pmc_summary.json
echo "pmc $TAG done"
