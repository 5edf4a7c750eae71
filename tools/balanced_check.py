#!/usr/bin/env python3
"""Does memory from mvhp_balanced_alloc_many() hold data (copy in, copy out), and what does the probe see?  (MVHP_BALANCED_TRACE=1)"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivideo_amd.hotpath import lib
L = lib()
L.mvhp_balanced_alloc_many.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
L.mvhp_balanced_free.argtypes = [C.c_int, C.c_void_p]
L.mvhp_balanced_info.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int * 4), C.POINTER(C.c_size_t)]
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
sizes = [int(3.3 * 2**30), int(12.1 * 2**30)]
arr, ptrs, g = (C.c_size_t * 2)(*sizes), (C.c_void_p * 2)(), C.c_int()
t0 = time.perf_counter()
rc = L.mvhp_balanced_alloc_many(0, 2, arr, ptrs, C.byref(g))
print("rc", rc, "groups", g.value, "%.2f s" % (time.perf_counter() - t0), flush=True)
for p, nb in zip(ptrs, sizes):
    per, ch = (C.c_int * 4)(), C.c_size_t()
    L.mvhp_balanced_info(0, p, C.byref(per), C.byref(ch))
    print(hex(p), nb, list(per), flush=True)
    src = torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev)
    back = torch.empty_like(src)
    assert hip.hipMemcpy(p, src.data_ptr(), nb, 3) == 0
    assert hip.hipMemcpy(back.data_ptr(), p, nb, 3) == 0
    torch.cuda.synchronize(dev)
    print("  copy in / out equal:", bool(torch.equal(src, back)), flush=True)
    del src, back
for p in ptrs:
    assert L.mvhp_balanced_free(0, p) == 1
