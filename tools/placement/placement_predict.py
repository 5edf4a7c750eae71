#!/usr/bin/env python3
"""Measurement: do the groups found by the pairwise write probe predict the decode launch's time?  One process: classify every
4 GB of a 230-GB allocation (mvhp_probe_pair, greedy clustering), then time the Baseline launch with its buffers at many
positions and print the groups under each buffer next to the time.
usage (GPU box, repo root): python tools/placement/placement_predict.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd import HotPath
from minivideo_amd.hotpath import lib
from minivideo_amd.synth import synth_packed

L = lib()
L.mvhp_probe_pair.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
dev = torch.device("cuda", 0)
GB = 1 << 30
A, STEP, W = 230, 4, 1 << 30
arena = torch.empty(A * GB, dtype=torch.uint8, device=dev)


def probe(a, b):   # GB offsets
    ms = C.c_float()
    assert L.mvhp_probe_pair(0, arena.data_ptr() + int(a * GB), arena.data_ptr() + int(b * GB), W, 2, C.byref(ms)) == 1
    return ms.value


# calibration: a window against its neighbour one GB on (same region almost surely) = "same group"
t_same = sorted(probe(x, x + 1) for x in (2, 34, 70, 100, 130, 170, 205))[3]
print("same-group pair time %.3f ms" % t_same, flush=True)
reps, group = [], {}
for x in range(0, A - 1, STEP):
    ts = [probe(r, x) if r != x else t_same for r in reps]
    best = max(range(len(ts)), key=lambda k: ts[k]) if ts else -1
    if best >= 0 and ts[best] >= 0.955 * t_same:
        group[x] = best
    else:
        group[x] = len(reps)
        reps.append(x)
print("groups per %d GB: %s" % (STEP, " ".join("%d:%s" % (x, "ABCDEFGH"[g]) for x, g in sorted(group.items()))), flush=True)


def groups_of(start_gb, size_gb):
    xs = sorted({min(group, key=lambda g: abs(g - x)) for x in [start_gb + 0.5 + k for k in range(int(size_gb + 0.999))]})
    hist = {}
    for x in [start_gb + 0.5 + k for k in range(int(size_gb + 0.999))]:
        g = group[int(x // STEP) * STEP]
        hist[g] = hist.get(g, 0) + 1
    return "".join("%s%d" % ("ABCDEFGH"[g], n) for g, n in sorted(hist.items()))


F = 2048
params, rec = synth_packed(120, 68, 16, seed=1000, profile="baseline", density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
pb, yb, rb = F * params.packed_bytes, F * params.yuv_bytes, F * params.rgb_bytes
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


def run(op, oy, orr):
    for _ in range(2):
        hot.recon_stages_dev(params, arena.data_ptr() + op * GB, F, arena.data_ptr() + oy * GB, arena.data_ptr() + orr * GB, sp, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(4):
        hot.recon_stages_dev(params, arena.data_ptr() + op * GB, F, arena.data_ptr() + oy * GB, arena.data_ptr() + orr * GB, sp, 3)
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / 4


def put_packed(op):
    arena[op * GB:op * GB + pb].view(F, -1).copy_(small.repeat(F // 16, 1))
    torch.cuda.synchronize(dev)


op = A - 14
put_packed(op)
print("packed at %d GB (%s); planes at X, RGB at X + 7" % (op, groups_of(op, pb / GB)), flush=True)
for x in range(0, A - 14 - 19, 8):
    print("  X = %3d: planes %-8s RGB %-10s %.3f ms" % (x, groups_of(x, yb / GB), groups_of(x + 7, rb / GB), run(op, x, x + 7)), flush=True)
print("planes at 0 (%s), RGB at X" % groups_of(0, yb / GB), flush=True)
for x in range(8, A - 14 - 12, 12):
    print("  X = %3d: RGB %-10s %.3f ms" % (x, groups_of(x, rb / GB), run(op, 0, x)), flush=True)
oy, orr = 40, 100
print("planes at %d (%s), RGB at %d (%s); packed at X" % (oy, groups_of(oy, yb / GB), orr, groups_of(orr, rb / GB)), flush=True)
for x in list(range(0, 28, 12)) + list(range(48, 88, 12)) + list(range(114, A - 13, 12)):
    put_packed(x)
    print("  X = %3d: packed %-10s %.3f ms" % (x, groups_of(x, pb / GB), run(x, oy, orr)), flush=True)
hot.close()
