#!/bin/bash
# Where a one-thumbnail run's time goes OUTSIDE the library's calls: the process and the HIP runtime.
# usage (GPU box, repo root): bash tools/startup_probe.sh > gpurun_out/<tag>_startup_probe.log
R=${GRAFT_REPO_ROOT:-$PWD}
T=$(mktemp -d /dev/shm/mvprobe_XXXX)
cd $T
python3 - <<PY
import subprocess, time, os, sys
def med(cmd, n=5, **kw):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True, text=True, **kw); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2], r
exe = "$R/minivideo_amd/mini_thumbnailer"
t, r = med(["/bin/true"]); print("fork+exec /bin/true                         %.3f s" % t)
t, r = med([exe]); print("mini_thumbnailer, no arguments (loads libminivideo.so + libamdhip64.so, no HIP call)  %.3f s" % t)
t, r = med([exe], env=dict(os.environ, LD_BIND_NOW="1")); print("mini_thumbnailer no arguments, LD_BIND_NOW  %.3f s" % t)
PY
# one thumbnail, the library's own account + the wall around the process
python3 - <<PY
import subprocess, time, os, sys
sys.path.insert(0, "$R")
import numpy as np
import bench
from minivideo_amd import gen
stream, _ = gen.make_stream(120, 68, 16, seed=1000, profile="baseline", dense=True, want_packed=False)
open("clip.264", "wb").write(bench.repeat_stream(stream, 16, 32).tobytes())
exe = "$R/minivideo_amd/mini_thumbnailer"
for envx in ({}, {"MINIVIDEO_FULL_EXIT": "1"}, {"HSA_ENABLE_SDMA": "0"}):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-i", "clip.264", "-f", "yuv420", "-n", "1"], capture_output=True, text=True, env=dict(os.environ, MINIVIDEO_STATS="1", **envx))
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("one thumbnail %s: median %.3f s" % (envx, ts[2]))
    for l in r.stderr.splitlines():
        if l.startswith("[minivideo]") or l.startswith("[mini_thumbnailer]"):
            print("    " + l[:230])
PY
ldd $R/minivideo_amd/libminivideo.so | wc -l
ls -la $R/minivideo_amd/libminivideo.so /opt/rocm/lib/libamdhip64.so* | awk '{print $5, $9}'
rm -rf $T
