#!/usr/bin/env python3
"""Wall time of mini_thumbnailer as a fresh process for SMALL jobs (1, 5, 20, 100 thumbnails out of a 100-picture 1080p stream on
tmpfs), yuv420 and the default png: what a user of the CLI sees -- process start and the HIP runtime's start-up included.
GPU box; prints one line per job (median of three runs)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from minivideo_amd import gen  # noqa: E402

stream, _ = gen.make_stream(120, 68, 16, seed=1000, profile="baseline", dense=True, want_packed=False)
d = tempfile.mkdtemp(prefix="mvsmall_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    path = os.path.join(d, "clip.264")
    bench.repeat_stream(stream, 16, 100).tofile(path)
    exe = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")
    for fmt in ("yuv420", "png"):
        for n in (1, 5, 20, 100):
            walls, last = [], ""
            for rep in range(3):
                for f in os.listdir(d):
                    if not f.endswith(".264"):
                        os.unlink(os.path.join(d, f))
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-i", path, "-f", fmt, "-n", str(n)], cwd=d, capture_output=True, text=True,
                                   env=dict(os.environ, MINIVIDEO_STATS="1"))
                walls.append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr
                last = [l for l in r.stderr.splitlines() if l.startswith("[minivideo] decode call:")][-1]
            n_files = len([f for f in os.listdir(d) if not f.endswith(".264")])
            print("%-6s -n %-3d  median %.3f s (%s)  files %d  | %s" % (fmt, n, sorted(walls)[1], " ".join("%.3f" % w for w in walls), n_files, last), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
