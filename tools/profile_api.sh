#!/bin/bash
# rocprofv3 kernel trace of the PUBLIC API path: the stock-interface CLI (minivideo_open / parse / decode) on a generated
# Annex-B file, so that the summary shows which reconstruction kernels minivideo_decode() launches and how many.
# usage (on the GPU box, from the repo root): bash tools/profile_api.sh <tag> <pictures, at most 999: the CLI limit> [profile]
# outputs: gpurun_out/<tag>_api_kernel_stats.csv, gpurun_out/<tag>_api.log (the library's own MINIVIDEO_STATS line + wall time)
set -e
TAG=$1; N=${2:-999}; PROFILE=${3:-baseline}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
W=$(mktemp -d /dev/shm/mvapi.XXXXXX)
trap 'rm -rf $W' EXIT
cd $R
python3 - "$W/clip.264" "$N" "$PROFILE" <<'PY'
import sys
sys.path.insert(0, ".")
from minivideo_amd import gen
from bench import repeat_stream
path, n, profile = sys.argv[1], int(sys.argv[2]), sys.argv[3]
s, _ = gen.make_stream(120, 68, 16, seed=1000, profile=profile, dense=True, want_packed=False)
repeat_stream(s, 16, n).tofile(path)
PY
export TMPDIR=/tmp MINIVIDEO_STATS=1
mkdir -p $OUT/prof_${TAG}_api
cd $W
t0=$(date +%s.%N)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_api -- $R/minivideo_amd/mini_thumbnailer -i $W/clip.264 -f yuv420 -n $N > $OUT/${TAG}_api.log 2>&1
t1=$(date +%s.%N)
echo "pictures $N files $(ls $W | grep -c yuv) wall_with_profiler_s $(python3 -c "print($t1 - $t0)")" >> $OUT/${TAG}_api.log
f=$(find $OUT/prof_${TAG}_api -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_api_kernel_stats.csv
grep -E "recon_|decode:" $OUT/${TAG}_api_kernel_stats.csv $OUT/${TAG}_api.log | cut -c1-200
