#!/usr/bin/env python3
"""Measurement: which half of the memory system is every 4 GB of one large allocation in?  (mvhp_probe_pair against offset 0)
usage (GPU box, repo root): python tools/placement/probe_map.py [GB of the arena]"""
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
A = int(sys.argv[1]) if len(sys.argv) > 1 else 230
arena = torch.empty(A * GB, dtype=torch.uint8, device=dev)
W = 256 << 20


def probe(a, b):
    ms = C.c_float()
    assert L.mvhp_probe_pair(0, arena.data_ptr() + a, arena.data_ptr() + b, W, 3, C.byref(ms)) == 1
    return ms.value


print("same window twice: %.3f ms; 0 and 256 MB: %.3f ms" % (probe(0, 0), probe(0, W)), flush=True)
line = []
for x in range(0, A, 2):
    t = probe(0, x * GB) if x else probe(0, W)
    line.append("%3d:%.2f" % (x, t))
    if len(line) == 8:
        print("  ".join(line), flush=True)
        line = []
if line:
    print("  ".join(line), flush=True)
