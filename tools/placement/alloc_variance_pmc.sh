#!/bin/bash
# tools/placement/alloc_variance.py under rocprofv3 --pmc, one counter group per pass: per-dispatch counters next to the script's own
# per-allocation times (1 warm + 2 timed launches per line of the log, in dispatch order).
# usage (GPU box, repo root): bash tools/placement/alloc_variance_pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/av_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $R/tools/placement/alloc_variance.py --launches 2 --warm 1 --trials 8 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed or timed out"; exit 1; }
  python3 - $OUT/pass$i $OUT/pass$i.log <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, log = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "recon_oct" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
lines = [l.rstrip() for l in open(log) if " ms" in l]
for k, l in enumerate(lines):
    mine = ids[3 * k + 1: 3 * k + 3]
    if not mine:
        break
    names = sorted(rows[mine[0]])
    print(l[:24], " ".join("%s=%.4g" % (n, sum(rows[m][n] for m in mine) / len(mine)) for n in names))
PY
done
