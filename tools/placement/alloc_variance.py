#!/usr/bin/env python3
"""Measurement: does the Baseline kernel's time depend on WHERE its buffers were allocated?  One process, the same
records, the output (and then the input) buffers re-allocated several times with the earlier ones kept alive, 10 launches each.
usage (GPU box, repo root): python tools/placement/alloc_variance.py [--profile baseline|high] [--frames 2048]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="baseline")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--trials", type=int, default=6)
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--warm", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
F = args.frames
params, rec = synth_packed(120, 68, 16, seed=1000, profile=args.profile, density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


def run(d_packed, d_yuv, d_rgb):
    torch.cuda.synchronize(dev)   # the records are copied on torch's stream, the kernels run on `st`
    n = args.launches
    for _ in range(args.warm):
        hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3)
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / n


def fresh(nbytes):
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


hold = []
d_packed = small.repeat(F // 16, 1).contiguous()
d_yuv, d_rgb = fresh(F * params.yuv_bytes), fresh(F * params.rgb_bytes)
print("first        %.3f ms  yuv %#x rgb %#x packed %#x" % (run(d_packed, d_yuv, d_rgb), d_yuv.data_ptr(), d_rgb.data_ptr(), d_packed.data_ptr()), flush=True)
print("again        %.3f ms" % run(d_packed, d_yuv, d_rgb), flush=True)
for t in range(args.trials):
    hold += [d_yuv, d_rgb]
    d_yuv, d_rgb = fresh(F * params.yuv_bytes), fresh(F * params.rgb_bytes)
    print("new outputs  %.3f ms  yuv %#x rgb %#x" % (run(d_packed, d_yuv, d_rgb), d_yuv.data_ptr(), d_rgb.data_ptr()), flush=True)
for t in range(3):
    hold.append(d_packed)
    d_packed = small.repeat(F // 16, 1).contiguous()
    print("new input    %.3f ms  packed %#x" % (run(d_packed, d_yuv, d_rgb), d_packed.data_ptr()), flush=True)
# the very first buffers again: is it the buffer or the moment?
print("first again  %.3f ms" % run(hold[-1] if False else d_packed, hold[0], hold[1]), flush=True)
# shifted views of one larger buffer: same physical pages, different offsets
big_y, big_r = fresh(F * params.yuv_bytes + (1 << 22)), fresh(F * params.rgb_bytes + (1 << 22))
for off in (0, 256, 4096, 65536, 1 << 20, 1 << 21):
    print("offset %-8d %.3f ms" % (off, run(d_packed, big_y[off:], big_r[off:])), flush=True)
hot.close()
