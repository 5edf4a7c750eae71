"""Shared helpers for tests: stream parsing through the C-ABI."""
import ctypes as C

import numpy as np

from minivideo_amd.hotpath import StreamParams, lib


class Stream:
    def __init__(self, data, spec=False):
        self.L = lib()
        self.L.mvhp_stream_last_error.restype = C.c_char_p
        self.L.mvhp_stream_open_ex.restype = C.c_int
        self.L.mvhp_stream_open_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.h = C.c_void_p()
        self.ok = self.L.mvhp_stream_open_ex(self.data.ctypes.data, self.data.size, 1 if spec else 0, C.byref(self.h)) == 1

    def close(self):
        if self.h:
            self.L.mvhp_stream_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def idr_count(self):
        return self.L.mvhp_stream_idr_count(self.h) if self.ok else 0

    def params(self, idr=0):
        p = StreamParams()
        rc = self.L.mvhp_stream_params(self.h, idr, C.byref(p))
        return p if rc == 1 else None

    def packed(self, idr):
        p = self.params(idr)
        if p is None:
            return 0, None
        out = np.zeros(p.packed_bytes, np.uint8)
        rc = self.L.mvhp_stream_decode_packed(self.h, idr, out.ctypes.data, out.size)
        return rc, out

    def error(self):
        e = self.L.mvhp_stream_last_error()
        return e.decode() if e else ""
