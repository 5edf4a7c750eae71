"""GPU: BASELINE.json's configurations at their real geometry, through the driver's own `pytest -m gpu` run.

* config 3 -- 120x68 High profile, CABAC, 8x8 transform: generated stream -> host front end (mvhp_stream_decode_packed) ->
  the three kernel layouts vs the oracle, and the same stream through the decode engine (what minivideo_decode runs).
  Exercises h264_transform.c:1205-1383 (8x8 residual) and h264_intra_prediction.c:1107-1793 (Intra8x8) at full-HD width:
  line buffers of 120 macroblocks, strips of four macroblocks across the whole row, 68 rows over 8 / 16 waves.
* config 5's one-GPU shares -- 512 pictures (N = 1) and 64 pictures (N = 8) of 1080p Baseline, 16 distinct pictures tiled:
  through mvhp_recon_batch_dev on the automatic layout (asserting which kernel pick_layout took) and through the engine.
* the launch bench.py times -- 2048 full-HD pictures on the automatic layout (eight pictures per workgroup), and 2080 (a
  ragged tail: pick_layout's round model prefers the banded four-picture form there).

Bit-exactness is checked against oracle/recon_ref.c on sampled pictures (every distinct source picture at least once for
the device-pointer launches)."""
import numpy as np
import pytest

from minivideo_amd import Engine, HotPath, gen
from minivideo_amd.hotpath import StreamParams
from oracle import loader
from tests.util import Stream

pytestmark = pytest.mark.gpu

W, H = 120, 68


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.fixture(scope="module")
def high1080():
    """two 1080p High/CABAC pictures: stream, the generator's records, the front end's records"""
    stream, packed = gen.make_stream(W, H, 2, seed=3003, profile="high")
    with Stream(stream) as s:
        assert s.ok and s.idr_count == 2
        p = s.params(0)
        recs = []
        for k in range(2):
            rc, r = s.packed(k)
            assert rc == 1, s.error()
            recs.append(r.reshape(W * H, 800))
    return stream, packed, p, np.stack(recs)


@pytest.fixture(scope="module")
def base1080():
    """sixteen distinct 1080p Baseline/CAVLC pictures (the bench's tiling) + their oracle pictures"""
    stream, packed = gen.make_stream(W, H, 16, seed=5005, profile="baseline")
    p = StreamParams(W, H, 0, 0, 0)
    ref = [loader.recon(p, packed[k], 1, want_rgb=True) for k in range(16)]
    return stream, packed, p, ref


def test_config3_front_end_equals_generator(high1080):
    _, packed, p, recs = high1080
    assert (p.width_mbs, p.height_mbs) == (W, H)
    assert np.array_equal(recs, packed)
    kinds = set(np.unique(packed[..., 0]).tolist())
    assert len(kinds) >= 3, kinds    # Intra4x4, Intra8x8 and Intra16x16 macroblocks are all present


def test_config3_high_cabac_1080p_kernels(hot, high1080):
    """`hot` runs this once per layout: one / four / eight pictures per workgroup"""
    _, _, p, recs = high1080
    yuv_g, rgb_g = hot.recon_host(p, recs, 2, want_rgb=True)
    yuv_o, rgb_o = loader.recon(p, recs, 2, want_rgb=True)
    assert np.array_equal(yuv_g, yuv_o)
    assert np.array_equal(rgb_g, rgb_o)


def test_config3_high_cabac_1080p_engine(high1080):
    stream, packed, p, _ = high1080
    got = {}

    def sink(seq, idr, rc, err, pr, yuv, rgb):
        got[seq] = (rc, yuv.copy() if yuv is not None else None, rgb.copy() if rgb is not None else None)
        return 1 if rc == 1 else 0

    eng = Engine(contexts=1)
    with Stream(stream) as s:
        rc, st = eng.decode(s.h, [0, 1, 1, 0, 1], want_rgb=True, sink=sink)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == 5
    for seq, idr in enumerate([0, 1, 1, 0, 1]):
        yuv_o, rgb_o = loader.recon(p, packed[idr], 1, want_rgb=True)
        assert got[seq][0] == 1 and np.array_equal(got[seq][1], yuv_o) and np.array_equal(got[seq][2], rgb_o), seq


def _tile_on_device(torch, packed, n):
    d_small = torch.from_numpy(packed.reshape(packed.shape[0], -1)).cuda()
    reps = (n + d_small.shape[0] - 1) // d_small.shape[0]
    d = d_small.repeat(reps, 1)[:n].contiguous()
    del d_small
    return d


# pictures per launch -> the kernel pick_layout takes on a 256-CU MI355X for Baseline pictures (hotpath_abi.hip: three waves per
# row of ONE picture up to 18 x CUs row-waves = 67 pictures, of four pictures up to 1.15 x CUs pictures, the plain banded form up to
# 0.84 x 4 x CUs, then one workgroup per four pictures up to a full round of 4 x CUs, beyond that whichever of the banded form (linear in the
# pictures), rounds of four and rounds of eight per CU the round model makes the shortest)
@pytest.mark.parametrize("n,layout", [(1, "pipe"), (2, "pipe1"), (40, "pipe1"), (64, "pipe1"), (128, "pipe"), (512, "quad_wide"), (1024, "quad"), (1100, "quad_wide"), (2048, "oct"), (2080, "quad_wide")])
def test_full_hd_batches_on_the_automatic_layout(torch_cuda, base1080, n, layout):
    torch = torch_cuda
    _, packed, p, ref = base1080
    d_packed = _tile_on_device(torch, packed, n)
    d_yuv = torch.zeros(n * p.yuv_bytes, dtype=torch.uint8, device="cuda")
    d_rgb = torch.zeros(n * p.rgb_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    hot = HotPath(0)
    try:
        hot.set_layout("auto")
        hot.recon_dev(p, d_packed.data_ptr(), n, d_yuv.data_ptr(), d_rgb.data_ptr(), st.cuda_stream)
        hot.sync_check(st.cuda_stream)
        took, waves = hot.last_launch()
        if torch.cuda.get_device_properties(0).multi_processor_count == 256:
            assert took == layout, (took, waves)
    finally:
        hot.close()
    yuv = d_yuv.view(n, -1)
    rgb = d_rgb.view(n, -1)
    # every distinct picture once at the front, once at the back, and a sample in between
    sample = sorted(set(range(min(16, n))) | set(range(max(0, n - 16), n)) | set(range(0, n, 131)))
    assert len(sample) >= min(8, n)
    for f in sample:
        yuv_o, rgb_o = ref[f % 16]
        assert np.array_equal(yuv[f].cpu().numpy(), yuv_o), f
        assert np.array_equal(rgb[f].cpu().numpy(), rgb_o), f
    del d_packed, d_yuv, d_rgb
    torch.cuda.empty_cache()


def _repeat_stream(stream, distinct, total):
    from bench import repeat_stream
    return repeat_stream(stream, distinct, total)


@pytest.mark.parametrize("n", [64, 512])
def test_config5_share_through_the_engine(base1080, n):
    """config 5 (512 x 1080p over N GPUs): the share one GPU gets at N = 8 (64) and at N = 1 (512), stream bytes -> pictures"""
    stream, packed, p, ref = base1080
    big = _repeat_stream(stream, 16, n)
    kept = {}
    sample = set(range(16)) | set(range(n - 16, n)) | set(range(0, n, 37))

    def sink(seq, idr, rc, err, pr, yuv, rgb):
        if rc == 1 and seq in sample:
            kept[seq] = (yuv.copy(), rgb.copy())
        return 1 if rc == 1 else 0

    eng = Engine(contexts=1)
    with Stream(big) as s:
        assert s.ok and s.idr_count == n
        rc, st = eng.decode(s.h, list(range(n)), want_rgb=True, sink=sink)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == n and st["pictures_failed"] == 0
    assert len(kept) == len(sample) >= 8
    for seq, (yuv, rgb) in kept.items():
        assert np.array_equal(yuv, ref[seq % 16][0]), seq
        assert np.array_equal(rgb, ref[seq % 16][1]), seq
