"""ctypes binding of include/minivideo_hotpath.h (the hot-path C-ABI)."""
import ctypes as C
import os

import numpy as np

SUCCESS, FAILURE, UNSUPPORTED = 1, 0, -1
MB_BYTES = 800
MB_HEADER_BYTES = 32

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MiniVideoError(RuntimeError):
    pass


class StreamParams(C.Structure):
    """mvhp_stream_params_t"""
    _fields_ = [
        ("width_mbs", C.c_uint32),
        ("height_mbs", C.c_uint32),
        ("chroma_qp_index_offset", C.c_int32),
        ("second_chroma_qp_index_offset", C.c_int32),
        ("flags", C.c_uint32),
    ]

    @property
    def mbs(self):
        return int(self.width_mbs) * int(self.height_mbs)

    @property
    def packed_bytes(self):
        return self.mbs * MB_BYTES

    @property
    def yuv_bytes(self):
        return self.mbs * 384

    @property
    def rgb_bytes(self):
        return self.mbs * 768


def lib_path():
    # MINIVIDEO_LIB: developer override used for A/B experiments with alternative builds of the same library
    return os.environ.get("MINIVIDEO_LIB") or os.path.join(_HERE, "libminivideo.so")


def lib():
    """Load libminivideo.so (in-tree build). Fails loudly when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise MiniVideoError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no pure-Python or CPU fallback for the reconstruction path)")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    pp = C.POINTER(StreamParams)
    L.mvhp_last_error.restype = C.c_char_p
    for f in ("mvhp_packed_frame_bytes", "mvhp_yuv_frame_bytes", "mvhp_rgb_frame_bytes"):
        getattr(L, f).restype = sz
        getattr(L, f).argtypes = [pp]
    L.mvhp_device_count.restype = i32
    L.mvhp_create.restype = i32
    L.mvhp_create.argtypes = [i32, C.POINTER(vp)]
    L.mvhp_destroy.restype = None
    L.mvhp_destroy.argtypes = [vp]
    L.mvhp_set_waves_per_picture.restype = i32
    L.mvhp_set_waves_per_picture.argtypes = [vp, i32]
    L.mvhp_set_layout.restype = i32
    L.mvhp_set_layout.argtypes = [vp, i32]
    L.mvhp_set_fused_color.restype = i32
    L.mvhp_set_fused_color.argtypes = [vp, i32]
    L.mvhp_recon_batch_dev.restype = i32
    L.mvhp_recon_batch_dev.argtypes = [vp, pp, vp, i32, vp, vp, vp]
    L.mvhp_recon_stages_dev.restype = i32
    L.mvhp_recon_stages_dev.argtypes = [vp, pp, vp, i32, vp, vp, vp, i32]
    L.mvhp_recon_batch_host.restype = i32
    L.mvhp_recon_batch_host.argtypes = [vp, pp, vp, i32, vp, vp]
    L.mvhp_sync_check.restype = i32
    L.mvhp_sync_check.argtypes = [vp, vp]
    L.mvhp_time_recon.restype = i32
    L.mvhp_time_recon.argtypes = [vp, pp, vp, i32, vp, vp, vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if hasattr(L, "mvhp_stream_open"):
        L.mvhp_stream_open.restype = i32
        L.mvhp_stream_open.argtypes = [vp, sz, C.POINTER(vp)]
        L.mvhp_stream_close.restype = None
        L.mvhp_stream_close.argtypes = [vp]
        L.mvhp_stream_idr_count.restype = i32
        L.mvhp_stream_idr_count.argtypes = [vp]
        L.mvhp_stream_params.restype = i32
        L.mvhp_stream_params.argtypes = [vp, i32, pp]
        L.mvhp_stream_decode_packed.restype = i32
        L.mvhp_stream_decode_packed.argtypes = [vp, i32, vp, sz]
    _LIB = L
    return L


def _err(L, what):
    msg = L.mvhp_last_error()
    return MiniVideoError(f"{what}: {msg.decode() if msg else 'failed'}")


class HotPath:
    """One GPU reconstruction context (mvhp_ctx_t): device + stream + scratch."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        if self._L.mvhp_create(int(device), C.byref(h)) != SUCCESS:
            raise _err(self._L, "mvhp_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.mvhp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_waves_per_picture(self, waves):
        if self._L.mvhp_set_waves_per_picture(self._h, int(waves)) != SUCCESS:
            raise MiniVideoError("waves per picture must be 0 (auto), 4, 8 or 16")

    def set_layout(self, layout):
        """0 auto, 1 one picture per workgroup (rows), 2 four pictures per workgroup (quad); speed only."""
        code = {"auto": 0, "rows": 1, "quad": 2, "oct": 3}.get(layout, layout)
        if self._L.mvhp_set_layout(self._h, int(code)) != SUCCESS:
            raise ValueError("layout must be auto/rows/quad/oct")

    def set_fused_color(self, on):
        self._L.mvhp_set_fused_color(self._h, 1 if on else 0)

    # -- host buffers ------------------------------------------------------
    def recon_host(self, params, packed, n_frames, want_rgb=False):
        """packed: uint8 array of n_frames*params.packed_bytes -> (yuv, rgb|None) uint8 arrays."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
        if packed.size != n_frames * params.packed_bytes:
            raise ValueError("packed buffer size does not match params/n_frames")
        yuv = np.empty(n_frames * params.yuv_bytes, dtype=np.uint8)
        rgb = np.empty(n_frames * params.rgb_bytes, dtype=np.uint8) if want_rgb else None
        rc = self._L.mvhp_recon_batch_host(
            self._h, C.byref(params), packed.ctypes.data, int(n_frames), yuv.ctypes.data,
            rgb.ctypes.data if want_rgb else None)
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_recon_batch_host")
        return yuv, rgb

    # -- device pointers (e.g. torch tensors' data_ptr()) --------------------
    def recon_dev(self, params, d_packed, n_frames, d_yuv, d_rgb=None, stream=None):
        rc = self._L.mvhp_recon_batch_dev(self._h, C.byref(params), d_packed, int(n_frames), d_yuv, d_rgb, stream)
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_recon_batch_dev")

    def recon_stages_dev(self, params, d_packed, n_frames, d_yuv, d_rgb, stream, stages):
        rc = self._L.mvhp_recon_stages_dev(self._h, C.byref(params), d_packed, int(n_frames), d_yuv, d_rgb,
                                           stream, int(stages))
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_recon_stages_dev")

    def sync_check(self, stream=None):
        if self._L.mvhp_sync_check(self._h, stream) != SUCCESS:
            raise _err(self._L, "mvhp_sync_check")

    def time_recon(self, params, d_packed, n_frames, d_yuv, d_rgb=None, stream=None, iters=10):
        a, b = C.c_float(0), C.c_float(0)
        rc = self._L.mvhp_time_recon(self._h, C.byref(params), d_packed, int(n_frames), d_yuv, d_rgb, stream,
                                     int(iters), C.byref(a), C.byref(b))
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_time_recon")
        return a.value, b.value
