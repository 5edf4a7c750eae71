#!/bin/bash
# like sweep.sh but each variant is "ENVVAR=value ... -- bench args"
COMMON=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  envpart=${v%%--*}; argpart=${v#*--}
  echo "## $envpart :: $COMMON $argpart" >> gpurun_out/sweep.log
  env $envpart timeout -k 10 240 python bench.py $COMMON $argpart 2>/dev/null | tail -1 >> gpurun_out/sweep.log || { echo "FAILED: $v" >> gpurun_out/sweep.log; }
done
python3 - <<'PY'
import json
for l in open('gpurun_out/sweep.log'):
    if l.startswith('##'): print(l.strip()); continue
    try:
        d=json.loads(l); print("   value %.4g MB/s  ms/step %.3f  kernel_ms %.3f  frac %.3f ok=%s" % (d['value'], d['ms_per_step'], [v for k,v in d['kernel_ms'].items() if k.startswith('recon')][0], d['roofline']['frac'], d['config']['bit_exact_vs_oracle']))
    except Exception as e: print("   ?", l[:200])
PY
