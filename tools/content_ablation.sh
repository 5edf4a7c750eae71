#!/bin/bash
# Which macroblock kinds, and how much of the divergence between the eight pictures of a wavefront, cost what: the shipped
# kernels timed on crafted content (bench.py --source records --kinds / --distinct 1).  MEASUREMENT AID; none of these is a
# BASELINE.json configuration.  usage (GPU box, repo root): bash tools/content_ablation.sh > gpurun_out/content_ablation.log
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
Q="--no-cpu-baseline --e2e-pictures 0 --placement-trials 0 --steps 10 --warmup 2"
run() {
  name=$1; shift
  out=$(timeout -k 10 200 python3 bench.py $Q "$@" 2>/dev/null) || { echo "$name FAILED"; return; }
  python3 - "$name" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["kernel_ms"]
name = [n for n in k if k[n] > 0][0]
print("%-44s %-18s %8.3f ms  frac %.3f  ok=%s" % (sys.argv[1], name, k[name], d["roofline"]["frac"], d["config"]["bit_exact_vs_oracle"]), flush=True)
PY
}
run "baseline stream (config 2)"
run "baseline stream, planes only (--no-rgb)" --no-rgb
run "baseline records mixed 0.4 I16 / 0.6 I4" --source records
run "baseline records all I16x16" --source records --kinds 1,0
run "baseline records all I4x4" --source records --kinds 0,0
run "baseline records, ONE distinct picture" --source records --distinct 1
run "baseline records, ONE distinct, all I4x4" --source records --distinct 1 --kinds 0,0
run "baseline records, ONE distinct, all I16" --source records --distinct 1 --kinds 1,0
run "baseline light" --density light
run "high stream (config 3)" --profile high
run "high records mixed" --source records --profile high
run "high records all I8x8" --source records --profile high --kinds 0,1
run "high records 0.4 I16 / 0.6 I8x8" --source records --profile high --kinds 0.4,1
run "high records, ONE distinct picture" --source records --profile high --distinct 1
run "high records ONE distinct all I8x8" --source records --profile high --distinct 1 --kinds 0,1
run "high stream on quad" --profile high --layout quad
run "baseline stream on quad" --layout quad
