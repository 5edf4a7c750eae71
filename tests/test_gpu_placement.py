"""mvhp_placed_alloc / mvhp_probe_pair (include/minivideo_hotpath.h): buffers of a batch placed inside one arena."""
import ctypes as C

import numpy as np
import pytest

from minivideo_amd import HotPath, PlacedBuffers, MiniVideoError, lib
from minivideo_amd.synth import synth_packed
from oracle import loader

pytestmark = pytest.mark.gpu
GB = 1 << 30


def _hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return h


def test_placed_buffers_are_disjoint_and_hold_data():
    sizes = [5 * GB, 1 * GB + 12345, 3 * GB]
    pb = PlacedBuffers(0, sizes, arena_bytes=32 * GB)
    try:
        assert len(pb.ptrs) == 3 and pb.groups_found >= 1 and all(p % (2 << 20) == 0 for p in pb.ptrs)
        spans = sorted((p, p + s) for p, s in zip(pb.ptrs, sizes))
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(2)), spans
        hip = _hip()
        rng = np.random.default_rng(5)
        for p, s in zip(pb.ptrs, sizes):
            tail = rng.integers(0, 256, 1 << 20, dtype=np.uint8)       # the last MiB of the buffer
            back = np.zeros_like(tail)
            assert hip.hipMemcpy(p + s - tail.size, tail.ctypes.data, tail.size, 1) == 0
            assert hip.hipMemcpy(back.ctypes.data, p + s - tail.size, tail.size, 2) == 0
            assert np.array_equal(tail, back)
    finally:
        pb.close()


def test_placed_alloc_refuses_an_arena_that_is_too_small():
    with pytest.raises(MiniVideoError):
        PlacedBuffers(0, [5 * GB, 5 * GB], arena_bytes=8 * GB)


def test_reconstruction_on_placed_buffers_matches_the_oracle():
    W, H, F = 20, 12, 24
    params, rec = synth_packed(W, H, F, seed=77, profile="high", density="dense")
    ref_yuv, ref_rgb = loader.recon(params, rec, F, want_rgb=True)
    pb = PlacedBuffers(0, [rec.size, F * params.yuv_bytes, F * params.rgb_bytes], arena_bytes=16 * GB)
    hot = HotPath(0)
    try:
        hip = _hip()
        flat = np.ascontiguousarray(rec).reshape(-1)
        assert hip.hipMemcpy(pb.ptrs[0], flat.ctypes.data, flat.size, 1) == 0
        for layout in ("rows", "quad", "oct"):
            hot.set_layout(layout)
            hot.recon_dev(params, pb.ptrs[0], F, pb.ptrs[1], pb.ptrs[2])
            hot.sync_check(None)
            yuv, rgb = np.empty(F * params.yuv_bytes, np.uint8), np.empty(F * params.rgb_bytes, np.uint8)
            assert hip.hipMemcpy(yuv.ctypes.data, pb.ptrs[1], yuv.size, 2) == 0
            assert hip.hipMemcpy(rgb.ctypes.data, pb.ptrs[2], rgb.size, 2) == 0
            assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb), layout
    finally:
        hot.close()
        pb.close()


def test_probe_pair_times_two_windows():
    L = lib()
    L.mvhp_probe_pair.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    pb = PlacedBuffers(0, [1 * GB, 1 * GB], arena_bytes=8 * GB)
    try:
        ms = C.c_float()
        assert L.mvhp_probe_pair(0, pb.ptrs[0], pb.ptrs[1], 256 << 20, 2, C.byref(ms)) == 1 and 0.0 < ms.value < 50.0
        assert L.mvhp_probe_pair(0, None, pb.ptrs[1], 256 << 20, 2, C.byref(ms)) == 0
    finally:
        pb.close()
