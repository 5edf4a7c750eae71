"""ctypes loader for the CPU checker (oracle/recon_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under minivideo_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_recon.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_recon.so"])
        L = C.CDLL(path)
        L.orc_recon_frame.restype = C.c_int
        L.orc_recon_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_yuv_to_rgb.restype = None
        L.orc_yuv_to_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_recon_batch.restype = C.c_int
        L.orc_recon_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def recon(params, packed, n_frames, want_rgb=False):
    """params: any ctypes struct laid out as mvhp_stream_params_t."""
    L = lib()
    mbs = int(params.width_mbs) * int(params.height_mbs)
    packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    assert packed.size == n_frames * mbs * 800
    yuv = np.zeros(n_frames * mbs * 384, dtype=np.uint8)
    rgb = np.zeros(n_frames * mbs * 768, dtype=np.uint8) if want_rgb else None
    ok = L.orc_recon_batch(C.byref(params), packed.ctypes.data, int(n_frames), yuv.ctypes.data,
                           rgb.ctypes.data if want_rgb else None)
    if ok != 1:
        raise RuntimeError("oracle: unsupported macroblock kind in packed buffer")
    return yuv, rgb
