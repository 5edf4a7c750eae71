"""ctypes binding of libmvgen.so, the synthetic H.264 IDR stream generator.

TEST / BENCH INFRASTRUCTURE: produces legal Annex-B streams of random syntax
elements plus the packed records a correct front end must derive from them."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class GenCfg(C.Structure):
    _fields_ = [
        ("width_mbs", C.c_int32), ("height_mbs", C.c_int32), ("n_frames", C.c_int32),
        ("seed", C.c_uint64),
        ("profile_idc", C.c_int32), ("cabac", C.c_int32), ("transform8x8", C.c_int32), ("dense", C.c_int32),
        ("cqp_offset_cb", C.c_int32), ("cqp_offset_cr", C.c_int32),
        ("sps_pps_every_frame", C.c_int32), ("allow_qp36_i16", C.c_int32),
        ("qp_min", C.c_int32), ("qp_max", C.c_int32), ("max_level", C.c_int32),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmvgen.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.mvgen_stream.restype = C.c_size_t
        L.mvgen_stream.argtypes = [C.POINTER(GenCfg), C.c_void_p, C.c_size_t, C.c_void_p]
        L.mvgen_stream_ex.restype = C.c_size_t
        L.mvgen_stream_ex.argtypes = [C.POINTER(GenCfg), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p]
        _LIB = L
    return _LIB


def make_stream(width_mbs, height_mbs, n_frames, seed=1, profile="baseline", dense=True, cqp_offsets=(0, 0),
                sps_pps_every_frame=False, qp_range=(24, 32), max_level=32, allow_qp36_i16=False, want_packed=True):
    """profile: 'baseline' (66, CAVLC), 'main' (77, CABAC), 'high' (100, CABAC, 8x8),
    'high_cavlc' (100, CAVLC, 8x8).  Returns (stream bytes as uint8 array, packed[n, W*H, 800] or None)."""
    prof = {"baseline": (66, 0, 0), "main": (77, 1, 0), "main_cavlc": (77, 0, 0), "high": (100, 1, 1),
            "high_cavlc": (100, 0, 1), "high_4x4": (100, 1, 0)}[profile]
    cfg = GenCfg(width_mbs, height_mbs, n_frames, seed, prof[0], prof[1], prof[2], 1 if dense else 0,
                 cqp_offsets[0], cqp_offsets[1], int(sps_pps_every_frame), int(allow_qp36_i16),
                 qp_range[0], qp_range[1], max_level)
    L = lib()
    packed = np.zeros((n_frames, width_mbs * height_mbs, 800), np.uint8) if want_packed else None
    n = L.mvgen_stream(C.byref(cfg), None, 0, packed.ctypes.data if want_packed else None)
    if n == 0:
        raise ValueError("generator rejected the configuration")
    out = np.zeros(n, np.uint8)
    n2 = L.mvgen_stream(C.byref(cfg), out.ctypes.data, n, packed.ctypes.data if want_packed else None)
    assert n2 == n
    return out, packed


def make_stream_ex(width_mbs, height_mbs, n_frames, seed=1, profile="baseline", slices=1, pcm_permille=0, scaling=0, dense=True,
                   cqp_offsets=(0, 0), sps_pps_every_frame=False, qp_range=(24, 32), max_level=32, allow_qp36_i16=True):
    """Streams OUTSIDE the reference's envelope (what MVHP_STREAM_SPEC decodes by the standard): `slices` slices
    per picture, pcm_permille / 1000 of the macroblocks I_PCM, scaling lists in the SPS (scaling & 1) and / or the PPS (scaling & 2;
    'high*' profiles only).  Returns (stream, packed[n, W*H, 800], weights[112] = scaling4[3][16] | scaling8[64] in raster order)."""
    prof = {"baseline": (66, 0, 0), "main": (77, 1, 0), "main_cavlc": (77, 0, 0), "high": (100, 1, 1),
            "high_cavlc": (100, 0, 1), "high_4x4": (100, 1, 0)}[profile]
    cfg = GenCfg(width_mbs, height_mbs, n_frames, seed, prof[0], prof[1], prof[2], 1 if dense else 0,
                 cqp_offsets[0], cqp_offsets[1], int(sps_pps_every_frame), int(allow_qp36_i16),
                 qp_range[0], qp_range[1], max_level)
    L = lib()
    packed = np.zeros((n_frames, width_mbs * height_mbs, 800), np.uint8)
    weights = np.zeros(112, np.uint8)
    n = L.mvgen_stream_ex(C.byref(cfg), slices, pcm_permille, scaling, None, 0, packed.ctypes.data, weights.ctypes.data)
    if n == 0:
        raise ValueError("generator rejected the configuration")
    out = np.zeros(n, np.uint8)
    n2 = L.mvgen_stream_ex(C.byref(cfg), slices, pcm_permille, scaling, out.ctypes.data, n, packed.ctypes.data, weights.ctypes.data)
    assert n2 == n
    return out, packed, weights
