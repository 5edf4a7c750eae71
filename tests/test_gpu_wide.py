"""GPU: the "wide" kernel forms -- one picture (one group of four) spread over several workgroups, bands of four macroblock rows on
different CUs, the rows between two bands handed over through global memory (recon_rows_kernel<.., WIDE> /
recon_quad_kernel<.., WIDE>; replaces the macroblock loop h264_slice.c:1046-1139 with its neighbour derivation
h264_spatial.c:333-416 for SMALL batches, where one workgroup per picture leaves most of the chip idle).

What is specific to these forms and therefore tested here (everything else runs through the `hot` fixture of conftest.py, which
asks every parity test for "wide" and "quad_wide" too):
* every size class at its real geometry, 1 .. 256 pictures, the layout asserted: 120x68 and 240x135, Baseline and High;
* band boundaries: heights that are / are not multiples of the band height, one-band pictures, widths 1 .. 5 (the hand-off
  asks for columns two ahead);
* the bookkeeping between launches of one context: ticket base, epoch tags, a seam buffer that grows, launches alternating
  between streams and between the two forms;
* hand-offs under uneven load: the same launches while another stream keeps the memory system busy, every byte checked.
Bit-exactness is against oracle/recon_ref.c."""
import numpy as np
import pytest

from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader

pytestmark = pytest.mark.gpu

WIDE = ("wide", "quad_wide", "pipe", "pipe1")


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


def _oracle(params, rec, distinct):
    return [loader.recon(params, rec[k], 1, want_rgb=True) for k in range(distinct)]


def _tile(torch, rec, n):
    d_small = torch.from_numpy(rec.reshape(rec.shape[0], -1)).cuda()
    reps = (n + d_small.shape[0] - 1) // d_small.shape[0]
    return d_small.repeat(reps, 1)[:n].contiguous()


def _launch_and_check(torch, hot, params, d_packed, n, ref, stream, layout, every=1):
    d_yuv = torch.zeros(n * params.yuv_bytes, dtype=torch.uint8, device="cuda")
    d_rgb = torch.zeros(n * params.rgb_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    hot.recon_dev(params, d_packed.data_ptr(), n, d_yuv.data_ptr(), d_rgb.data_ptr(), stream.cuda_stream)
    hot.sync_check(stream.cuda_stream)
    assert hot.last_launch()[0] == layout, hot.last_launch()
    yuv, rgb = d_yuv.view(n, -1).cpu().numpy(), d_rgb.view(n, -1).cpu().numpy()
    D = len(ref)
    for f in range(0, n, every):
        assert np.array_equal(yuv[f], ref[f % D][0]), (layout, n, f)
        assert np.array_equal(rgb[f], ref[f % D][1]), (layout, n, f)


@pytest.mark.parametrize("layout", WIDE)
@pytest.mark.parametrize("profile", ["baseline", "high"])
def test_full_hd_1_to_256_pictures(torch_cuda, layout, profile):
    """120x68: 17 bands per picture; 1, 2, 7 pictures (short groups of four), 64 (config 5's share of a GPU at N = 8), 256"""
    torch = torch_cuda
    params, rec = synth_packed(120, 68, 8, seed=404, profile=profile, density="dense")
    ref = _oracle(params, rec, 8)
    st = torch.cuda.Stream()
    hot = HotPath(0)
    try:
        hot.set_layout(layout)
        for n in (1, 2, 7, 64, 256):
            d_packed = _tile(torch, rec, n)
            _launch_and_check(torch, hot, params, d_packed, n, ref, st, layout, every=1 if n <= 64 else 5)
            del d_packed
    finally:
        hot.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layout", WIDE)
def test_2160p_high(torch_cuda, layout):
    """240x135: 34 bands, the last of three rows; line buffers of 240 macroblocks"""
    torch = torch_cuda
    params, rec = synth_packed(240, 135, 3, seed=2161, profile="high", density="dense")
    ref = _oracle(params, rec, 3)
    st = torch.cuda.Stream()
    hot = HotPath(0)
    try:
        hot.set_layout(layout)
        for n in (1, 5, 16):
            d_packed = _tile(torch, rec, n)
            _launch_and_check(torch, hot, params, d_packed, n, ref, st, layout)
            del d_packed
    finally:
        hot.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layout", WIDE)
@pytest.mark.parametrize("W,H", [(1, 4), (1, 5), (2, 8), (3, 9), (4, 12), (5, 13), (9, 3), (31, 16), (33, 17), (2, 41)])
def test_band_boundaries(layout, W, H):
    """heights around multiples of the band height (4; 8 when asked for), widths below the hand-off's look-ahead"""
    hot = HotPath(0)
    try:
        hot.set_layout(layout)
        for waves in (0, 8):
            hot.set_waves_per_picture(waves)
            for n in (1, 3, 6):
                params, rec = synth_packed(W, H, n, seed=W * 1000 + H * 10 + n, profile="high", density="dense", qp_range=(10, 45))
                yuv_g, rgb_g = hot.recon_host(params, rec, n, want_rgb=True)
                yuv_o, rgb_o = loader.recon(params, rec, n, want_rgb=True)
                assert np.array_equal(yuv_g, yuv_o) and np.array_equal(rgb_g, rgb_o), (layout, W, H, waves, n)
                assert hot.last_launch()[0] == layout
    finally:
        hot.close()


def test_automatic_choice_by_batch_size(torch_cuda):
    """pick_layout on a 256-CU device.  Baseline: one picture four times per wavefront with three waves
    per row; up to 18 x CUs row-waves (pictures x rows) one picture per wavefront with three waves per row; up to 76 x CUs
    row-waves (and 2 x CUs pictures) four per wavefront again; up to 0.84 x 4 x CUs pictures four pictures in bands; then one workgroup per group.
    Batches that may hold Intra8x8 macroblocks: up to 46 x CUs row-waves one picture per wavefront with three waves per row, up
    to 76 x CUs row-waves one picture in bands, then as Baseline."""
    torch = torch_cuda
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("thresholds are stated for 256 CUs")
    hot = HotPath(0)
    try:
        hot.set_layout("auto")
        for (W, H, n, flags, want) in [(20, 17, 1, 0, "pipe"), (20, 17, 2, 0, "pipe1"), (20, 17, 3, 1, "pipe1"), (20, 17, 4, 1, "pipe1"), (20, 17, 4, 0, "pipe1"),
                                       (20, 17, 271, 0, "pipe1"), (20, 17, 272, 0, "pipe"),
                                       (20, 17, 512, 0, "pipe"), (20, 17, 513, 0, "quad_wide"), (20, 68, 67, 0, "pipe1"), (20, 68, 68, 0, "pipe"),
                                       (20, 68, 286, 0, "pipe"),
                                       (20, 68, 287, 0, "quad_wide"), (20, 68, 173, 1, "pipe1"), (20, 68, 174, 1, "wide"),
                                       (20, 68, 286, 1, "wide"), (20, 68, 287, 1, "quad_wide"),
                                       (6, 68, 860, 0, "quad_wide"), (6, 68, 861, 0, "quad")]:
            params, rec = synth_packed(W, H, 4, seed=7, profile="baseline", density="light")
            params.flags = flags   # bit 0: MVHP_PARAM_MAY_HAVE_8X8 (a hint for this choice only)
            d_packed = _tile(torch, rec, n)
            d_yuv = torch.empty(n * params.yuv_bytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            hot.recon_dev(params, d_packed.data_ptr(), n, d_yuv.data_ptr(), None, None)
            hot.sync_check(None)
            assert hot.last_launch()[0] == want, (W, H, n, hot.last_launch())
            yuv_o, _ = loader.recon(params, rec[(n - 1) % 4], 1)
            assert np.array_equal(d_yuv.view(n, -1)[n - 1].cpu().numpy(), yuv_o)
            del d_packed, d_yuv
    finally:
        hot.close()
    torch.cuda.empty_cache()


def test_automatic_choice_on_odd_shapes(torch_cuda):
    """whatever pick_layout takes for pictures of one row, one column, two by two, 720p-like and portrait shapes at batch sizes on
    both sides of every threshold: the sampled pictures (first, last, four in between) are the oracle's, and the launch reports no
    error word"""
    torch = torch_cuda
    hot = HotPath(0)
    seen = set()
    try:
        hot.set_layout("auto")
        for (W, H) in [(2, 2), (20, 1), (1, 20), (30, 17), (9, 45)]:
            for profile in ("baseline", "high"):
                params, rec = synth_packed(W, H, 4, seed=W * 100 + H, profile=profile, density="dense")
                ref = _oracle(params, rec, 4)
                for n in (1, 3, 5, 68, 100, 300, 320, 321, 400, 861, 1100):
                    d_packed = _tile(torch, rec, n)
                    d_yuv = torch.zeros(n * params.yuv_bytes, dtype=torch.uint8, device="cuda")
                    d_rgb = torch.zeros(n * params.rgb_bytes, dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    hot.recon_dev(params, d_packed.data_ptr(), n, d_yuv.data_ptr(), d_rgb.data_ptr(), None)
                    hot.sync_check(None)
                    seen.add(hot.last_launch()[0])
                    yuv = d_yuv.view(n, -1)
                    rgb = d_rgb.view(n, -1)
                    for f in sorted({0, n - 1, n // 2, n // 3, (2 * n) // 3, max(0, n - 2)}):
                        assert np.array_equal(yuv[f].cpu().numpy(), ref[f % 4][0].reshape(-1)), (W, H, profile, n, f, hot.last_launch())
                        assert np.array_equal(rgb[f].cpu().numpy(), ref[f % 4][1].reshape(-1)), (W, H, profile, n, f, hot.last_launch())
                    del d_packed, d_yuv, d_rgb
    finally:
        hot.close()
    torch.cuda.empty_cache()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert {"pipe", "pipe1", "wide", "quad_wide", "quad"} <= seen, seen


def test_bookkeeping_across_launches(torch_cuda):
    """one context, sixty launches: both forms in turn, batch sizes up and down (the seam buffer grows, later launches find
    tags of earlier ones in it), two streams in turn (a launch waits for the context's previous wide launch on the other
    stream: they share the ticket counter and the seams)"""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    shapes = {}
    for (W, H) in [(11, 9), (20, 17), (7, 35)]:
        params, rec = synth_packed(W, H, 6, seed=W + H, profile="high", density="dense")
        shapes[(W, H)] = (params, rec, _oracle(params, rec, 6))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    hot = HotPath(0)
    pending = []
    try:
        for it in range(60):
            W, H = list(shapes)[int(rng.integers(0, 3))]
            params, rec, ref = shapes[(W, H)]
            n = int(rng.choice([1, 3, 4, 9, 30, 70]))
            layout = WIDE[it % 4]
            hot.set_layout(layout)
            st = streams[int(rng.integers(0, 2))]
            d_packed = _tile(torch, rec, n)
            d_yuv = torch.zeros(n * params.yuv_bytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # (allocations and the tiling above ran on torch's default stream)
            hot.recon_dev(params, d_packed.data_ptr(), n, d_yuv.data_ptr(), None, st.cuda_stream)
            assert hot.last_launch()[0] == layout
            pending.append((d_packed, d_yuv, n, ref, st))
            if len(pending) == 4:   # several launches in flight on two streams before anything is checked
                for (dp, dy, nn, rf, s) in pending:
                    hot.sync_check(s.cuda_stream)
                    yuv = dy.view(nn, -1).cpu().numpy()
                    for f in range(nn):
                        assert np.array_equal(yuv[f], rf[f % 6][0]), (it, nn, f)
                pending = []
    finally:
        hot.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layout", WIDE)
def test_hand_offs_under_uneven_load(torch_cuda, layout):
    """the seams are written with write-through stores and polled with agent-scope loads, no fence anywhere: run the launches
    while a second stream streams copies through HBM and L2 (uneven load shifts which band runs ahead), check every byte"""
    torch = torch_cuda
    params, rec = synth_packed(60, 34, 8, seed=909, profile="high", density="dense")
    ref = _oracle(params, rec, 8)
    st, noise = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    big2 = torch.empty_like(big)
    hot = HotPath(0)
    try:
        hot.set_layout(layout)
        for n in (3, 40, 17, 96, 5, 64):
            d_packed = _tile(torch, rec, n)
            d_yuv = torch.zeros(n * params.yuv_bytes, dtype=torch.uint8, device="cuda")
            d_rgb = torch.zeros(n * params.rgb_bytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(noise):   # ~25 ms of copies; the reconstruction launches run inside that window
                for _ in range(12):
                    big2.copy_(big)
                    big.copy_(big2)
            for _ in range(3):
                hot.recon_dev(params, d_packed.data_ptr(), n, d_yuv.data_ptr(), d_rgb.data_ptr(), st.cuda_stream)
            hot.sync_check(st.cuda_stream)
            assert hot.last_launch()[0] == layout
            torch.cuda.synchronize()
            yuv, rgb = d_yuv.view(n, -1).cpu().numpy(), d_rgb.view(n, -1).cpu().numpy()
            for f in range(n):
                assert np.array_equal(yuv[f], ref[f % 8][0]) and np.array_equal(rgb[f], ref[f % 8][1]), (layout, n, f)
            del d_packed, d_yuv, d_rgb
    finally:
        hot.close()
    del big, big2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layout", WIDE)
@pytest.mark.parametrize("delta", [-1, 1])
def test_a_lost_unit_ends_the_launch_with_an_error(layout, delta):
    """the waits between workgroups are bounded and any failure in a launch ends every wait: hand the units out one off (test
    hook) -- with -1 the first band of the first picture is never reconstructed and the band below it waits for a seam that never
    comes, with +1 a ticket falls outside the launch -- and the launch must END with the error word set, mvhp_recon_batch_host
    must fail, and the context must reconstruct correctly again afterwards"""
    import ctypes as C
    import time
    from minivideo_amd import MiniVideoError
    hot = HotPath(0)
    try:
        hot.set_layout(layout)
        params, rec = synth_packed(20, 17, 5, seed=31, profile="high", density="dense")
        yuv_o, _ = loader.recon(params, rec, 5)
        yuv, _ = hot.recon_host(params, rec, 5)
        assert np.array_equal(yuv, yuv_o) and hot.last_launch()[0] == layout
        hot._L.mvhp_debug_skew_next_ticket_base.argtypes = [C.c_void_p, C.c_int]
        assert hot._L.mvhp_debug_skew_next_ticket_base(hot._h, delta) == 1
        t0 = time.perf_counter()
        with pytest.raises(MiniVideoError, match="error word"):
            hot.recon_host(params, rec, 5)
        assert time.perf_counter() - t0 < 30.0
        yuv, _ = hot.recon_host(params, rec, 5)     # the hook was for one launch; the counter's bookkeeping is intact
        assert np.array_equal(yuv, yuv_o)
    finally:
        hot.close()
