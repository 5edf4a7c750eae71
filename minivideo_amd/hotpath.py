"""ctypes binding of include/minivideo_hotpath.h (the hot-path C-ABI)."""
import ctypes as C
import os

import numpy as np

SUCCESS, FAILURE, UNSUPPORTED = 1, 0, -1
MB_BYTES = 800
MB_HEADER_BYTES = 32

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


LAYOUTS = ("auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1")   # MVHP_LAYOUT_*


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
        ("scaling4", (C.c_uint8 * 16) * 3),   # MVHP_PARAM_SCALING: weight matrices, raster order (Intra Y, Cb, Cr)
        ("scaling8", C.c_uint8 * 64),         # ... Intra Y 8x8
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
    L.mvhp_expand_compact_dev.restype = i32
    L.mvhp_expand_compact_dev.argtypes = [vp, pp, vp, sz, i32, vp, vp]
    L.mvhp_recon_stages_dev.restype = i32
    L.mvhp_recon_stages_dev.argtypes = [vp, pp, vp, i32, vp, vp, vp, i32]
    L.mvhp_recon_batch_host.restype = i32
    L.mvhp_recon_batch_host.argtypes = [vp, pp, vp, i32, vp, vp]
    L.mvhp_sync_check.restype = i32
    L.mvhp_sync_check.argtypes = [vp, vp]
    L.mvhp_last_launch_info.restype = i32
    L.mvhp_last_launch_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
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
            raise MiniVideoError("waves per picture must be 0 (auto), 1, 2, 4, 6, 8, 12 or 16")

    def set_layout(self, layout):
        """0 auto, 1 one picture per workgroup (rows), 2 four pictures per workgroup (quad), 3 eight (oct), 4 one picture over
        several workgroups (wide), 5 four pictures over several workgroups (quad_wide); speed only."""
        code = LAYOUTS.index(layout) if layout in LAYOUTS else layout
        if self._L.mvhp_set_layout(self._h, int(code)) != SUCCESS:
            raise ValueError("layout must be one of " + "/".join(LAYOUTS))

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

    def expand_compact_dev(self, params, d_compact, stride, n_pictures, d_packed, stream=None):
        """compact pictures (transfer format) -> packed records, both in device memory"""
        rc = self._L.mvhp_expand_compact_dev(self._h, C.byref(params), d_compact, int(stride), int(n_pictures), d_packed, stream)
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_expand_compact_dev")

    def recon_stages_dev(self, params, d_packed, n_frames, d_yuv, d_rgb, stream, stages):
        rc = self._L.mvhp_recon_stages_dev(self._h, C.byref(params), d_packed, int(n_frames), d_yuv, d_rgb,
                                           stream, int(stages))
        if rc != SUCCESS:
            raise _err(self._L, "mvhp_recon_stages_dev")

    def sync_check(self, stream=None):
        if self._L.mvhp_sync_check(self._h, stream) != SUCCESS:
            raise _err(self._L, "mvhp_sync_check")

    def last_launch(self):
        """(layout name, waves per workgroup) of the last reconstruction launch -- speed-only choices of the launcher."""
        lay, nw = C.c_int(0), C.c_int(0)
        self._L.mvhp_last_launch_info(self._h, C.byref(lay), C.byref(nw))
        return (LAYOUTS[lay.value] if 0 <= lay.value < len(LAYOUTS) else "?"), nw.value


# ---------------------------------------------------------------------------
# placed batch buffers (mvhp_placed_alloc): records, planes and RGB each in a group of the memory system of its own
# ---------------------------------------------------------------------------
class PlacedBuffers:
    """Device buffers of the given sizes inside one large allocation, each -- as far as the device shows several groups of
    memory regions -- in a group of its own (DESIGN.md 3 "Placement"; the fastest placement of a batch's three streams).
    .ptrs: device addresses; .groups: group index per buffer (-1 = straddles); .groups_found; .seconds.  close() frees all."""

    def __init__(self, device, sizes, arena_bytes=0):
        import time
        L = lib()
        L.mvhp_placed_alloc.restype = C.c_int
        L.mvhp_placed_alloc.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mvhp_placed_free.restype = None
        L.mvhp_placed_free.argtypes = [C.c_void_p]
        n = len(sizes)
        arr, ptrs, arena = (C.c_size_t * n)(*[int(v) for v in sizes]), (C.c_void_p * n)(), C.c_void_p()
        gof, gf = (C.c_int * n)(), C.c_int()
        t0 = time.perf_counter()
        if L.mvhp_placed_alloc(int(device), n, arr, int(arena_bytes), ptrs, C.byref(arena), gof, C.byref(gf)) != SUCCESS:
            raise MiniVideoError("mvhp_placed_alloc: not enough free device memory for the arena (use ordinary allocations)")
        self.seconds = time.perf_counter() - t0
        self._L, self._arena = L, arena
        self.ptrs = [int(p) for p in ptrs]
        self.groups = [int(g) for g in gof]
        self.groups_found = int(gf.value)

    def close(self):
        if getattr(self, "_arena", None):
            self._L.mvhp_placed_free(self._arena)
            self._arena = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# decode engine (mvhp_engine_*): stream bytes -> reconstructed pictures, pipelined
# ---------------------------------------------------------------------------
class EngineOpts(C.Structure):
    """mvhp_engine_opts_t"""
    _fields_ = [("contexts", C.c_int32), ("host_threads", C.c_int32), ("batch_pictures", C.c_int32),
                ("chunk_pictures", C.c_int32), ("fail_context", C.c_int32), ("first_device", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class DecodeStats(C.Structure):
    """mvhp_decode_stats_t"""
    _fields_ = [("pictures_issued", C.c_uint32), ("pictures_ok", C.c_uint32), ("pictures_failed", C.c_uint32),
                ("batches", C.c_uint32), ("batches_requeued", C.c_uint32), ("contexts", C.c_uint32),
                ("host_threads", C.c_uint32), ("launches_by_layout", C.c_uint32 * 4), ("max_batch_pictures", C.c_uint32),
                ("wall_s", C.c_double), ("entropy_busy_s", C.c_double), ("h2d_s", C.c_double), ("kernel_s", C.c_double),
                ("d2h_s", C.c_double), ("sink_s", C.c_double), ("stream_bytes", C.c_uint64), ("h2d_bytes", C.c_uint64),
                ("d2h_bytes", C.c_uint64), ("host_alloc_s", C.c_double), ("dev_alloc_s", C.c_double),
                ("first_launch_s", C.c_double), ("first_picture_s", C.c_double), ("host_alloc_bytes", C.c_uint64),
                ("dev_alloc_bytes", C.c_uint64), ("placed_buffers", C.c_uint32), ("reserved", C.c_uint32),
                ("launches_wide", C.c_uint32 * 4)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


SINK_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(StreamParams),
                     C.POINTER(C.c_uint8), C.POINTER(C.c_uint8))


class Engine:
    """mvhp_engine_t: host entropy threads -> H2D -> batched kernels -> D2H -> sink, over every context."""

    def __init__(self, contexts=0, host_threads=0, batch_pictures=0, chunk_pictures=0, fail_context=-1, first_device=0, placed=False):
        self._L = L = lib()
        L.mvhp_engine_create.restype = C.c_int
        L.mvhp_engine_create.argtypes = [C.POINTER(EngineOpts), C.POINTER(C.c_void_p)]
        L.mvhp_engine_destroy.restype = None
        L.mvhp_engine_destroy.argtypes = [C.c_void_p]
        L.mvhp_engine_decode.restype = C.c_int
        L.mvhp_engine_decode.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, SINK_T,
                                         C.c_void_p, C.POINTER(DecodeStats)]
        L.mvhp_engine_release_picture.restype = None
        L.mvhp_engine_release_picture.argtypes = [C.c_void_p, C.c_int]
        o = EngineOpts(contexts, host_threads, batch_pictures, chunk_pictures, fail_context, first_device)
        o.reserved[0] = 1 if placed else 0   # bit 0: batch buffers from mvhp_placed_alloc_sets (same as MINIVIDEO_PLACED=1)
        h = C.c_void_p()
        if L.mvhp_engine_create(C.byref(o), C.byref(h)) != SUCCESS:
            raise MiniVideoError("mvhp_engine_create failed (no HIP device? there is no CPU reconstruction path)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mvhp_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_picture(self, seq):
        """gives back a picture whose sink call answered 2 (any thread; the decode call returns when the last one is back)"""
        self._L.mvhp_engine_release_picture(self._h, int(seq))

    def decode(self, stream_handle, order, wanted=None, want_rgb=False, sink=None):
        """sink(seq, idr, rc, err, params, yuv ndarray | None, rgb ndarray | None) -> 1 accept / 0 reject / -1 stop /
        2 accept and keep until release_picture(seq); the arrays are views of page-locked memory valid only during the call
        (or until the release).  Returns (rc, stats dict)."""
        order = (C.c_int * len(order))(*order)
        st = DecodeStats()

        def _cb(user, seq, idr, rc, err, p, yuv, rgb):
            if sink is None:
                return 1 if rc == SUCCESS else 0
            pr = p.contents
            y = np.ctypeslib.as_array(yuv, shape=(pr.yuv_bytes,)) if yuv else None
            r = np.ctypeslib.as_array(rgb, shape=(pr.rgb_bytes,)) if rgb else None
            return int(sink(seq, idr, rc, err.decode() if err else "", pr, y, r))

        cb = SINK_T(_cb) if sink is not None else C.cast(None, SINK_T)
        rc = self._L.mvhp_engine_decode(self._h, stream_handle, order, len(order), len(order) if wanted is None else wanted,
                                        int(want_rgb), cb, None, C.byref(st))   # 0 planes, 1 planes + RGB, 3 RGB only
        return rc, st.as_dict()
