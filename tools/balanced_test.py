#!/usr/bin/env python3
"""Measurement: the Baseline launch on buffers from mvhp_balanced_alloc() against ordinary allocations, one process.
usage (GPU box, repo root): python tools/balanced_test.py [--profile baseline|high]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivideo_amd import HotPath
from minivideo_amd.hotpath import lib
from minivideo_amd.synth import synth_packed
from oracle import loader

ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="baseline")
args = ap.parse_args()
L = lib()
L.mvhp_balanced_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
L.mvhp_balanced_free.argtypes = [C.c_int, C.c_void_p]
L.mvhp_balanced_info.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int * 4), C.POINTER(C.c_size_t)]
dev = torch.device("cuda", 0)
F = 2048
params, rec = synth_packed(120, 68, 16, seed=1000, profile=args.profile, density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
pb, yb, rb = F * params.packed_bytes, F * params.yuv_bytes, F * params.rgb_bytes
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream


class Raw:   # a device pointer as something torch.as_tensor() understands
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


L.mvhp_balanced_alloc_many.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def balanced_many(sizes, names):
    n = len(sizes)
    arr, ptrs, g = (C.c_size_t * n)(*sizes), (C.c_void_p * n)(), C.c_int()
    t0 = time.perf_counter()
    rc = L.mvhp_balanced_alloc_many(0, n, arr, ptrs, C.byref(g))
    dt = time.perf_counter() - t0
    if rc != 1:
        print("balanced allocation failed", flush=True)
        return None
    print("%d buffers in %.2f s, %d groups" % (n, dt, g.value), flush=True)
    for p, nb, what in zip(ptrs, sizes, names):
        per, ch = (C.c_int * 4)(), C.c_size_t()
        L.mvhp_balanced_info(0, p, C.byref(per), C.byref(ch))
        print("  %-7s %5.2f GB: chunks per group %s (chunk %d MB)" % (what, nb / 2**30, list(per), ch.value >> 20), flush=True)
    return [p for p in ptrs]


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


# ordinary allocations first (three sets)
d_packed = small.repeat(F // 16, 1).contiguous()
hold = []
for k in range(3):
    y, r = torch.empty(yb, dtype=torch.uint8, device=dev), torch.empty(rb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    hold += [y, r]
    print("ordinary allocation %d: %.3f ms" % (k, run(d_packed.data_ptr(), y.data_ptr(), r.data_ptr())), flush=True)
ref_y = hold[0].clone()
ref_r = hold[1].clone()
del hold, y, r
torch.cuda.empty_cache()
res = balanced_many([pb, yb, rb], ["packed", "planes", "RGB"])
if res:
    bp, by_, br = res
    assert hip.hipMemcpy(bp, d_packed.data_ptr(), pb, 3) == 0
    torch.cuda.synchronize(dev)
    print("all three balanced:        %.3f ms" % run(bp, by_, br), flush=True)
    print("all three balanced again:  %.3f ms" % run(bp, by_, br), flush=True)
    oy, orr = torch.empty(yb, dtype=torch.uint8, device=dev), torch.empty(rb, dtype=torch.uint8, device=dev)
    assert hip.hipMemcpy(oy.data_ptr(), by_, yb, 3) == 0 and hip.hipMemcpy(orr.data_ptr(), br, rb, 3) == 0
    print("same bytes as on the ordinary buffers:", bool(torch.equal(oy, ref_y)) and bool(torch.equal(orr, ref_r)), flush=True)
    del oy, orr
    print("outputs balanced, ordinary input: %.3f ms" % run(d_packed.data_ptr(), by_, br), flush=True)
    for p in (bp, by_, br):
        assert L.mvhp_balanced_free(0, p) == 1
    print("freed; free memory now %.1f GB" % (torch.cuda.mem_get_info(dev)[0] / 2**30), flush=True)
hot.close()
