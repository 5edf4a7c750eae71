"""PCIe-inclusive rate of mvhp_recon_batch_host(): packed records in page-locked host memory -> planes (+RGB) in
page-locked host memory, one call = H2D + kernel + D2H on the context's stream.  Not the bench.py metric."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from minivideo_amd import HotPath, lib
from minivideo_amd.synth import synth_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = lib()
L.mvhp_host_alloc.restype = C.c_void_p
L.mvhp_host_alloc.argtypes = [C.c_size_t]
L.mvhp_host_free.argtypes = [C.c_void_p]
L.mvhp_recon_batch_host.restype = C.c_int
L.mvhp_recon_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
params, rec = synth_packed(120, 68, 8, seed=3, profile="baseline", density="dense")
pb, yb, rb = params.packed_bytes * n, params.yuv_bytes * n, params.rgb_bytes * n
hp, hy, hr = L.mvhp_host_alloc(pb), L.mvhp_host_alloc(yb), L.mvhp_host_alloc(rb)
src = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(pb,))
flat = rec.reshape(-1)
for k in range(n):
    src[k * params.packed_bytes:(k + 1) * params.packed_bytes] = flat[(k % 8) * params.packed_bytes:((k % 8) + 1) * params.packed_bytes]
h = HotPath(0)
out = {}
for want_rgb in (False, True):
    for _ in range(2):
        assert L.mvhp_recon_batch_host(h._h, C.byref(params), hp, n, hy, hr if want_rgb else None) == 1
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        assert L.mvhp_recon_batch_host(h._h, C.byref(params), hp, n, hy, hr if want_rgb else None) == 1
    dt = (time.perf_counter() - t0) / reps
    out["rgb" if want_rgb else "yuv"] = {"pictures": n, "seconds": dt, "macroblocks_per_s": n * params.mbs / dt,
                                         "host_bytes_moved": pb + yb + (rb if want_rgb else 0),
                                         "GB_per_s": (pb + yb + (rb if want_rgb else 0)) / dt / 1e9}
print(json.dumps(out))
