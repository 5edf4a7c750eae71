#!/usr/bin/env python3
"""Measurement: the Baseline launch as a function of the ABSOLUTE position of its buffers inside one 230 GB allocation.
usage (GPU box, repo root): python tools/placement/placement_map.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed

dev = torch.device("cuda", 0)
F = 2048
params, rec = synth_packed(120, 68, 16, seed=1000, profile="baseline", density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
GB = 1 << 30
free, total = torch.cuda.mem_get_info(dev)
print("free %.1f GB of %.1f GB" % (free / GB, total / GB), flush=True)
A = 230
arena = torch.empty(A * GB, dtype=torch.uint8, device=dev)
pb, yb, rb = F * params.packed_bytes, F * params.yuv_bytes, F * params.rgb_bytes
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


def run(op, oy, orr):
    for _ in range(2):
        hot.recon_stages_dev(params, arena.data_ptr() + op, F, arena.data_ptr() + oy, arena.data_ptr() + orr, sp, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(4):
        hot.recon_stages_dev(params, arena.data_ptr() + op, F, arena.data_ptr() + oy, arena.data_ptr() + orr, sp, 3)
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / 4


def put_packed(op):
    arena[op:op + pb].view(F, -1).copy_(small.repeat(F // 16, 1))
    torch.cuda.synchronize(dev)


# 1. outputs swept, input parked at the far end
op = (A - 14) * GB
put_packed(op)
print("packed at %d GB; planes at X, RGB at X + 7 GB" % (op >> 30), flush=True)
for x in range(0, A - 14 - 19, 8):
    print("  X = %3d GB: %.3f ms" % (x, run(op, x * GB, (x + 7) * GB)), flush=True)
# 2. planes and RGB separately
print("packed at %d GB; planes at 0, RGB at X" % (op >> 30), flush=True)
for x in range(8, A - 14 - 12, 16):
    print("  X = %3d GB: %.3f ms" % (x, run(op, 0, x * GB)), flush=True)
print("packed at %d GB; RGB at 0, planes at X" % (op >> 30), flush=True)
for x in range(16, A - 14 - 6, 16):
    print("  X = %3d GB: %.3f ms" % (x, run(op, x * GB, 0)), flush=True)
# 3. input swept, outputs parked at the far end
oy, orr = (A - 20) * GB, (A - 13) * GB
print("planes at %d GB, RGB at %d GB; packed at X" % (oy >> 30, orr >> 30), flush=True)
for x in range(0, A - 20 - 13, 16):
    put_packed(x * GB)
    print("  X = %3d GB: %.3f ms" % (x, run(x * GB, oy, orr)), flush=True)
hot.close()
