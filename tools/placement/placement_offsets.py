#!/usr/bin/env python3
"""Measurement: the Baseline launch with its three buffers at controlled offsets inside ONE large allocation (physically
contiguous if the device memory is fresh): does the time depend on the distances between input, planes and RGB?
usage (GPU box, repo root): python tools/placement/placement_offsets.py"""
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
arena = torch.empty(100 * GB, dtype=torch.uint8, device=dev)
pb, yb, rb = F * params.packed_bytes, F * params.yuv_bytes, F * params.rgb_bytes
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


def al(x, a=1 << 21):
    return (x + a - 1) // a * a


def run(op, oy, orr):
    for _ in range(2):
        hot.recon_stages_dev(params, arena.data_ptr() + op, F, arena.data_ptr() + oy, arena.data_ptr() + orr, sp, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        hot.recon_stages_dev(params, arena.data_ptr() + op, F, arena.data_ptr() + oy, arena.data_ptr() + orr, sp, 3)
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / 5


def put_packed(op):
    v = arena[op:op + pb].view(F, -1)
    v.copy_(small.repeat(F // 16, 1))
    torch.cuda.synchronize(dev)


put_packed(0)
print("arena %#x; packed at 0 (%.2f GB), planes %.2f GB, RGB %.2f GB" % (arena.data_ptr(), pb / GB, yb / GB, rb / GB), flush=True)
base_y = al(pb)
for gy in (0, 64 << 20, 256 << 20, 1 * GB, 2 * GB, 3 * GB, 4 * GB, 6 * GB, 8 * GB, 16 * GB, 32 * GB):
    oy = base_y + gy
    orr = al(oy + yb)
    print("planes at packed_end + %6d MB, RGB right behind: %.3f ms" % (gy >> 20, run(0, oy, orr)), flush=True)
oy = base_y
for gr in (0, 64 << 20, 256 << 20, 1 * GB, 2 * GB, 3 * GB, 4 * GB, 6 * GB, 8 * GB, 16 * GB, 32 * GB):
    orr = al(oy + yb) + gr
    print("planes right behind packed, RGB at planes_end + %6d MB: %.3f ms" % (gr >> 20, run(0, oy, orr)), flush=True)
# input moved instead
oy, orr = 60 * GB, 70 * GB
for op in (0, 1 * GB, 2 * GB, 4 * GB, 8 * GB, 16 * GB, 32 * GB, 40 * GB):
    put_packed(op)
    print("outputs fixed at 60 / 70 GB, packed at %2d GB: %.3f ms" % (op >> 30, run(op, oy, orr)), flush=True)
hot.close()
