#!/usr/bin/env python3
"""Measurement: the launch on buffers from mvhp_placed_alloc() against ordinary allocations, one process.
usage (GPU box, repo root): python tools/placement/placed_check.py [--profile baseline|high] [--mbs 120x68] [--frames 2048]"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd import HotPath
from minivideo_amd.hotpath import lib
from minivideo_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="baseline")
ap.add_argument("--mbs", default="120x68")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--layout", default="auto")
args = ap.parse_args()
wm, hm = (int(v) for v in args.mbs.split("x"))
L = lib()
L.mvhp_placed_alloc.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_int), C.POINTER(C.c_int)]
L.mvhp_placed_free.argtypes = [C.c_void_p]
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dev = torch.device("cuda", 0)
F = args.frames
params, rec = synth_packed(wm, hm, 16, seed=1000, profile=args.profile, density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
pb, yb, rb = F * params.packed_bytes, F * params.yuv_bytes, F * params.rgb_bytes
hot = HotPath(0)
hot.set_fused_color(True)
hot.set_layout(args.layout)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


def run(pp, py, pr, n=5):
    for _ in range(2):
        hot.recon_stages_dev(params, pp, F, py, pr, sp, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        hot.recon_stages_dev(params, pp, F, py, pr, sp, 3)
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / n


d_packed = small.repeat(F // 16, 1).contiguous()
hold = []
for k in range(3):
    y, r = torch.empty(yb, dtype=torch.uint8, device=dev), torch.empty(rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    hold += [y, r]
    print("ordinary allocation %d: %.3f ms" % (k, run(d_packed.data_ptr(), y.data_ptr(), r.data_ptr())), flush=True)
ref_y, ref_r = hold[0], hold[1]
del hold[2:], y, r
torch.cuda.empty_cache()
sizes = (C.c_size_t * 3)(pb, yb, rb)
ptrs, arena, gof, gf = (C.c_void_p * 3)(), C.c_void_p(), (C.c_int * 3)(), C.c_int()
t0 = time.perf_counter()
rc = L.mvhp_placed_alloc(0, 3, sizes, 0, ptrs, C.byref(arena), gof, C.byref(gf))
print("mvhp_placed_alloc: rc %d in %.2f s, %d groups in the arena, records / planes / RGB in groups %s" % (rc, time.perf_counter() - t0, gf.value, list(gof)), flush=True)
if rc == 1:
    bp, by_, br = ptrs[0], ptrs[1], ptrs[2]
    assert hip.hipMemcpy(bp, d_packed.data_ptr(), pb, 3) == 0
    torch.cuda.synchronize(dev)
    print("placed buffers:        %.3f ms" % run(bp, by_, br), flush=True)
    print("placed buffers again:  %.3f ms" % run(bp, by_, br), flush=True)
    oy, orr = torch.empty(yb, dtype=torch.uint8, device=dev), torch.empty(rb, dtype=torch.uint8, device=dev)
    assert hip.hipMemcpy(oy.data_ptr(), by_, yb, 3) == 0 and hip.hipMemcpy(orr.data_ptr(), br, rb, 3) == 0
    torch.cuda.synchronize(dev)
    print("same bytes as on the ordinary buffers:", bool(torch.equal(oy, ref_y)), bool(torch.equal(orr, ref_r)), flush=True)
    L.mvhp_placed_free(arena)
hot.close()
