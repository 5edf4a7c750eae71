"""Reference expansion of the compact transfer format (include/minivideo_hotpath.h, "compact pictures") into packed
records -- numpy / Python, TEST INFRASTRUCTURE: the product expands on the GPU (mvhp_expand_compact_dev)."""
import ctypes as C

import numpy as np

from minivideo_amd.hotpath import lib

COMPACT_MB_BYTES_MAX = 804
COMPACT_SLACK_BYTES = 1536
COMPACT_MAX_ENTRIES = 191


def decode_compact(stream, idr):
    """-> (rc, bytes used, uint8 buffer) through mvhp_stream_decode_compact."""
    L = lib()
    L.mvhp_stream_decode_compact.restype = C.c_int
    L.mvhp_stream_decode_compact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    p = stream.params(idr)
    if p is None:
        return 0, 0, None
    buf = np.zeros(p.mbs * COMPACT_MB_BYTES_MAX + COMPACT_SLACK_BYTES, np.uint8)
    used = C.c_size_t(0)
    rc = L.mvhp_stream_decode_compact(stream.h, idr, buf.ctypes.data, buf.size, C.byref(used))
    return rc, int(used.value), buf


def expand(buf, mbs):
    """compact picture -> packed records [mbs, 800]"""
    out = np.zeros((mbs, 800), np.uint8)
    off = buf[:4 * mbs].view(np.uint32)
    base = 4 * mbs
    for mb in range(mbs):
        r = base + int(off[mb])
        hdr = buf[r:r + 32].copy()
        n = int(hdr[28:32].view(np.uint32)[0])
        dense = bool(hdr[5] & 1)
        hdr[28:32] = 0                                   # reserved1 and flags of a packed record are zero
        hdr[5] = 0
        out[mb, :32] = hdr
        if dense:
            out[mb, 32:] = buf[r + 32:r + 800]
            continue
        ent = buf[r + 32:r + 32 + 4 * n].view(np.uint32)
        coef = out[mb, 32:].view(np.int16)
        coef[(ent & 0xffff).astype(np.int64)] = (ent >> 16).astype(np.uint16).view(np.int16)
    return out
