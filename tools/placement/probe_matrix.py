#!/usr/bin/env python3
"""Measurement: pairwise mvhp_probe_pair between windows 8 GB apart in one large allocation -> which windows share a
part of the memory system?  usage (GPU box, repo root): python tools/placement/probe_matrix.py [window MB] [step GB]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd.hotpath import lib

L = lib()
L.mvhp_probe_pair.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
dev = torch.device("cuda", 0)
GB = 1 << 30
A = 232
W = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
arena = torch.empty(A * GB, dtype=torch.uint8, device=dev)
pos = list(range(0, A - 2, step))


def probe(a, b):
    ms = C.c_float()
    assert L.mvhp_probe_pair(0, arena.data_ptr() + a * GB, arena.data_ptr() + b * GB, W, 2, C.byref(ms)) == 1
    return ms.value


print("window %d MB; rows/columns = GB offsets; entry = ms per pass of writing both windows" % (W >> 20), flush=True)
print("      " + " ".join("%4d" % p for p in pos), flush=True)
for a in pos:
    row = []
    for b in pos:
        row.append("%4.2f" % probe(a, b if b != a else a))
    print("%4d: " % a + " ".join(row), flush=True)
