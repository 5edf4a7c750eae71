#!/usr/bin/env python3
"""Developer aid (GPU box): why bench.py's end-to-end calls run ~5 % slower than tools/e2e_ab.py's on the same box.  One process,
one engine, the same 2048-picture job with (a) no sink, (b) a trivial Python sink, (c) bench.py's sink (copies four pictures),
(d) as (c) with torch imported and its context synchronised before every call -- in turn, several rounds, medians."""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from minivideo_amd import Engine, gen, lib  # noqa: E402

n_distinct, total = 16, 2048
stream, _ = gen.make_stream(120, 68, n_distinct, seed=1000, profile="baseline", dense=True, want_packed=False)
big = bench.repeat_stream(stream, n_distinct, total)
L = lib()
h = C.c_void_p()
assert L.mvhp_stream_open(big.ctypes.data, big.size, C.byref(h)) == 1
order = list(range(total))
eng = Engine(contexts=1, host_threads=16)
eng.decode(h, order, want_rgb=True)
check = [0, 1, total // 2, total - 1]
kept = {}


def bench_sink(seq, idr, rc, err, p, yuv, rgb):
    if rc == 1 and seq in check:
        kept[seq] = (idr, yuv.copy(), rgb.copy() if rgb is not None else None)
    return 1 if rc == 1 else 0


variants = [("no sink", None, False), ("trivial sink", lambda *a: 1 if a[2] == 1 else 0, False), ("bench sink", bench_sink, False), ("bench sink + torch", bench_sink, True)]
walls = {v[0]: [] for v in variants}
torch = None
for r in range(6):
    for name, sink, use_torch in (variants if r % 2 == 0 else variants[::-1]):
        if use_torch:
            if torch is None:
                import torch
            torch.cuda.synchronize(torch.device("cuda", 0))
        t0 = time.perf_counter()
        rc, st = eng.decode(h, order, want_rgb=True, sink=sink)
        if use_torch:
            torch.cuda.synchronize(torch.device("cuda", 0))
        walls[name].append(time.perf_counter() - t0)
        assert rc == 1
for name, w in walls.items():
    print("%-20s median %.4f s  (%s)" % (name, statistics.median(w), " ".join("%.3f" % x for x in w)), flush=True)
