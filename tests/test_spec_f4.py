"""SURVEY 8f row f4, the remainder (round 3; opt-in MVHP_STREAM_SPEC / MINIVIDEO_SPEC=1, OUTSIDE the parity contract): pictures of
several slices, SPS / PPS scaling matrices, I_PCM macroblocks.  The reference decodes none of them correctly (one slice per
picture h264_slice.c:1019; scaling lists land in sps_array[0] without fall-back rules h264_parameterset.c:723-736, PPS lists
answer UNSUPPORTED :904-923; I_PCM answers UNSUPPORTED h264_macroblock.c:151-154), so the authority here is the standard:

* host front end against the generator's independent formulation (libmvgen.so mvgen_stream_ex: picture-wide maps with a slice
  id per macroblock, its own reading of Table 7-2's fall-back rules, its own I_PCM writer incl. the CABAC flush / restart);
* hand-derived samples for the reconstruction rules (derivations below), checked on the oracle (CPU) and on the kernels (GPU);
* HIP against the oracle on generated streams, through the record path and through the engine.

Hand derivations (Appendix-A macroblock: Intra16x16, DC prediction, one luma DC level +3, QP'Y 28, nothing else coded):
  f = 3 at all 16 positions of the DC matrix; qP = 28 < 36: dcY = (f * LevelScale(28 % 6 = 4, 0, 0) + 2^(5 - 4)) >> (6 - 4).
  Flat_4x4_16: LevelScale = 16 * 16 = 256 -> dcY = (768 + 2) >> 2 = 192; the block's only coefficient d00 = 192 -> every
  residual = (192 + 32) >> 6 = 3 -> Y = 128 + 3 = 131 (the reference's own output).
  * scaling: weight[Y][0][0] = 32 instead of 16 -> LevelScale = 32 * 16 = 512 (8.5.9: weightScale * normAdjust) -> dcY =
    (1536 + 2) >> 2 = 384 -> residual (384 + 32) >> 6 = 6 -> Y = 134.  Weight 8 -> 128 -> (384 + 2) >> 2 = 96 -> 2 -> Y = 130.
    Chroma: one Cb DC level +2 at QP'c 28 (chroma_qp_index_offset 0): f = 2 (all four); dcC = ((f * LS) << (28 / 6 = 4)) >> 5 =
    (2 * 256 * 16) >> 5 = 256 -> residual (256 + 32) >> 6 = 4 -> Cb = 132 with flat lists; weight[Cb][0][0] = 24 -> LS = 384 ->
    (2 * 384 * 16) >> 5 = 384 -> (384 + 32) >> 6 = 6 -> Cb = 134; Cr (weight 16, no level) stays 128.
  * slices: macroblock 1 = Intra16x16 DC without residual beside macroblock 0 (Y = 131).  Same slice: its left neighbour is
    available, no upper one -> DC = (sum of 16 left samples + 8) >> 4 = 131.  New slice starting at macroblock 1: the left
    neighbour belongs to another slice, hence not available (6.4.8) -> DC = 128 (8.3.3.3, no neighbour).
  * I_PCM: the samples are the picture (8.3.5); a macroblock predicted from them sees them as neighbours: vertical prediction
    below a PCM macroblock whose bottom row is 10, 20, .. repeats that row."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from minivideo_amd import gen
from minivideo_amd.hotpath import StreamParams
from oracle import loader
from tests.compact import decode_compact, expand
from tests.kat import kat_packed
from tests.util import Stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLICES, SCALING = 4, 8    # MVHP_PARAM_SLICES, MVHP_PARAM_SCALING


def _weights_of(p):
    return np.concatenate([np.frombuffer(bytes(p.scaling4), np.uint8), np.frombuffer(bytes(p.scaling8), np.uint8)])


def _flat_params(W, H, flags=0):
    p = StreamParams(W, H, 0, 0, flags)
    C.memset(C.byref(p, StreamParams.scaling4.offset), 16, 112)
    return p


CASES = [("baseline", 3, 0, 0), ("baseline", 1, 90, 0), ("main", 4, 60, 0), ("main_cavlc", 2, 0, 0), ("high", 1, 0, 1),
         ("high", 1, 0, 2), ("high", 3, 40, 3), ("high_cavlc", 5, 120, 3), ("high_4x4", 2, 30, 1), ("high", 6, 0, 0)]


@pytest.mark.parametrize("profile,slices,pcm,scaling", CASES)
def test_front_end_matches_the_generator(profile, slices, pcm, scaling):
    for seed in range(4):
        W, H = 3 + seed * 3, 2 + seed * 2
        stream, packed, weights = gen.make_stream_ex(W, H, 3, seed=seed * 11 + slices, profile=profile, slices=slices,
                                                     pcm_permille=pcm, scaling=scaling, qp_range=(0, 51) if seed == 3 else (24, 32))
        with Stream(stream, spec=True) as s:
            assert s.ok and s.idr_count == 3, s.error()
            for k in range(3):
                p = s.params(k)
                assert p is not None and bool(p.flags & SLICES) == (min(slices, W * H) > 1)
                nonflat = bool((weights != 16).any())
                assert bool(p.flags & SCALING) == (scaling != 0 and nonflat)
                if p.flags & SCALING:
                    assert np.array_equal(_weights_of(p), weights)
                rc, rec = s.packed(k)
                assert rc == 1, s.error()
                assert np.array_equal(rec.reshape(-1, 800), packed[k]), (seed, k)
                rc, used, buf = decode_compact(s, k)     # the transfer format carries the same records
                assert rc == 1 and np.array_equal(expand(buf, W * H), packed[k])
        if pcm:
            assert (packed[..., 0] == 3).any() or W * H < 12
        if slices > 1 and W > 1:
            assert (packed[..., 6] != 0).any()


def test_reference_mode_refuses_what_the_reference_cannot_decode():
    """default mode = the reference's envelope: scaling lists and I_PCM are refused, a further slice is a picture of its own"""
    stream, _, _ = gen.make_stream_ex(4, 3, 1, seed=2, profile="high", scaling=1)
    with Stream(stream) as s:
        assert s.idr_count == 1 and s.params(0) is None            # the SPS was refused (UNSUPPORTED)
    stream, packed, _ = gen.make_stream_ex(4, 3, 1, seed=3, profile="baseline", pcm_permille=1000)
    with Stream(stream) as s:
        assert s.ok and s.packed(0)[0] != 1 and "I_PCM" in s.error()
    stream, _, _ = gen.make_stream_ex(4, 3, 1, seed=4, profile="main", slices=2)
    with Stream(stream) as s:
        # h264_slice.c:96-99 exports after each slice NAL: two "pictures", the first ends early (documented divergence: the
        # build fails it instead of exporting a picture with holes, h264_slice.c:1047-1139), the second starts mid-picture
        assert s.idr_count == 2
        rc0, _ = s.packed(0)
        assert rc0 != 1 and "slice ends before the last macroblock" in s.error()


def test_early_ending_cavlc_slice_fails_the_picture_in_reference_mode():
    """CAVLC twin of the above: the reference's more_rbsp_data() stays true to the end of its sample (H12), it runs into the
    next NAL unit's bytes; the build reports the truncation"""
    stream, _, _ = gen.make_stream_ex(4, 3, 1, seed=5, profile="baseline", slices=2)
    with Stream(stream) as s:
        assert s.idr_count == 2 and s.packed(0)[0] != 1


# ---- hand vectors on the oracle (and, below, on the GPU) -------------------------------------------------------------------
def _kat_variants():
    """(name, params, records[2 macroblocks], expected Y of macroblock 0, of macroblock 1, expected Cb of macroblock 0)"""
    out = []
    _, rec = kat_packed(28)
    rec = rec.reshape(2, 800).copy()
    rec[1, 4] = 2                                   # macroblock 1: Intra16x16 DC prediction (header byte 4 = i16_pred_mode)
    rec[1, 3] = 0                                   # chroma DC prediction
    out.append(("flat, one slice", _flat_params(2, 1), rec.copy(), 131, 131, 128))
    r = rec.copy()
    r[1, 6] = 1                                     # MVHP_UNAVAIL_A
    out.append(("unavail bit without MVHP_PARAM_SLICES is ignored", _flat_params(2, 1), r.copy(), 131, 131, 128))
    out.append(("macroblock 1 starts a slice", _flat_params(2, 1, SLICES), r.copy(), 131, 128, 128))
    p = _flat_params(2, 1, SCALING)
    p.scaling4[0][0] = 32
    out.append(("luma DC weight 32", p, rec.copy(), 134, 134, 128))
    p = _flat_params(2, 1, SCALING)
    p.scaling4[0][0] = 8
    out.append(("luma DC weight 8", p, rec.copy(), 130, 130, 128))
    r = rec.copy()
    r[0, 32 + 2 * 256:32 + 2 * 256 + 2] = np.array([2], np.int16).view(np.uint8)     # Cb DC level +2 (slot 256)
    r[0, 8:12] = np.array([r[0, 8:12].view(np.uint32)[0] | (1 << 16)], np.uint32).view(np.uint8)   # nz_mask: Cb block 0
    out.append(("chroma DC, flat", _flat_params(2, 1), r.copy(), 131, 131, 132))
    p = _flat_params(2, 1, SCALING)
    p.scaling4[1][0] = 24
    out.append(("chroma DC weight 24", p, r.copy(), 131, 131, 134))
    return out


def test_a_picture_of_many_small_slices_passes_the_size_guard():
    """the size guard (a slice NAL must be able to hold the picture its SPS announces: >= 1 payload bit per 8 macroblocks) counts
    ALL the slices of a picture in spec mode: a full-HD picture of 68 one-row slices of flat content has slice NAL units far
    below the 128 bytes the whole picture needs; in reference mode, where one slice NAL is one picture, the same
    NAL units are refused one by one"""
    W, H = 120, 68
    stream, packed, _ = gen.make_stream_ex(W, H, 2, seed=9, profile="main", slices=H, dense=False)
    b = stream.tobytes()
    starts = [i for i in range(len(b) - 4) if b[i:i + 4] == b"\x00\x00\x00\x01"] + [len(b)]
    slice_sizes = [starts[k + 1] - starts[k] - 4 for k in range(len(starts) - 1) if (b[starts[k] + 4] & 31) == 5]
    assert len(slice_sizes) == 2 * H and min(slice_sizes) * 64 < W * H, (len(slice_sizes), min(slice_sizes))
    with Stream(stream, spec=True) as s:
        assert s.ok and s.idr_count == 2
        for k in range(2):
            rc, rec = s.packed(k)
            assert rc == 1, s.error()
            assert np.array_equal(rec.reshape(W * H, 800), packed[k])
    with Stream(stream) as s:   # reference mode: every slice NAL is a picture of its own, too small for 120 x 68 macroblocks
        if s.ok:
            assert all(s.packed(k)[0] != 1 for k in range(s.idr_count))


@pytest.mark.parametrize("case", _kat_variants(), ids=lambda c: c[0])
def test_hand_vectors_on_the_oracle(case):
    _, p, rec, y0, y1, cb0 = case
    yuv, _ = loader.recon(p, rec, 1)
    Y = yuv[:512].reshape(16, 32)
    assert np.all(Y[:, :16] == y0) and np.all(Y[:, 16:] == y1)
    assert np.all(yuv[512:640].reshape(8, 16)[:, :8] == cb0) and np.all(yuv[640:] == 128)


def _pcm_records():
    """2 x 2 macroblocks: (0,0) I_PCM with a ramp, (1,0) Intra16x16 horizontal from it, (0,1) Intra16x16 vertical from it,
    (1,1) I_PCM again (all samples 200)"""
    rec = np.zeros((4, 800), np.uint8)
    luma = (np.arange(256).reshape(16, 16) % 16 * 10 + np.arange(16)[:, None]).astype(np.uint8)     # row y: y, 10 + y, 20 + y ..
    cb = np.full((8, 8), 90, np.uint8) + np.arange(8, dtype=np.uint8)[None, :]
    cr = np.full((8, 8), 160, np.uint8) - np.arange(8, dtype=np.uint8)[:, None]

    def pcm(rec_mb, Y, Cb, Cr):
        rec_mb[0] = 3
        area = rec_mb[32:]
        for j in range(8):
            area[64 * j:64 * j + 16] = Y[2 * j]
            area[64 * j + 16:64 * j + 32] = Y[2 * j + 1]
            area[64 * j + 32:64 * j + 40] = Cb[j]
            area[64 * j + 40:64 * j + 48] = Cr[j]

    pcm(rec[0], luma, cb, cr)
    rec[1, 0], rec[1, 1], rec[1, 4], rec[1, 3] = 2, 28, 1, 1      # Intra16x16 horizontal, chroma horizontal
    rec[2, 0], rec[2, 1], rec[2, 4], rec[2, 3] = 2, 28, 0, 2      # Intra16x16 vertical, chroma vertical
    pcm(rec[3], np.full((16, 16), 200, np.uint8), np.full((8, 8), 7, np.uint8), np.full((8, 8), 250, np.uint8))
    return rec, luma, cb, cr


def _check_pcm_picture(yuv):
    rec, luma, cb, cr = _pcm_records()
    Y = yuv[:1024].reshape(32, 32)
    Cb = yuv[1024:1280].reshape(16, 16)
    Cr = yuv[1280:1536].reshape(16, 16)
    assert np.array_equal(Y[:16, :16], luma) and np.array_equal(Cb[:8, :8], cb) and np.array_equal(Cr[:8, :8], cr)
    assert np.array_equal(Y[:16, 16:], np.repeat(luma[:, 15:16], 16, axis=1))      # horizontal: the PCM block's right column
    assert np.array_equal(Y[16:, :16], np.repeat(luma[15:16, :], 16, axis=0))      # vertical: its bottom row
    assert np.array_equal(Cb[:8, 8:], np.repeat(cb[:, 7:8], 8, axis=1)) and np.array_equal(Cr[8:, :8], np.repeat(cr[7:8, :], 8, axis=0))
    assert np.all(Y[16:, 16:] == 200) and np.all(Cb[8:, 8:] == 7) and np.all(Cr[8:, 8:] == 250)


def test_pcm_hand_vector_on_the_oracle():
    rec, *_ = _pcm_records()
    yuv, _ = loader.recon(_flat_params(2, 2), rec, 1)
    _check_pcm_picture(yuv)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", _kat_variants(), ids=lambda c: c[0])
def test_hand_vectors_on_the_gpu(hot, case):
    _, p, rec, y0, y1, cb0 = case
    yuv, rgb = hot.recon_host(p, rec, 1, want_rgb=True)
    ref_yuv, ref_rgb = loader.recon(p, rec, 1, want_rgb=True)
    assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb)
    assert np.all(yuv[:512].reshape(16, 32)[:, :16] == y0) and np.all(yuv[:512].reshape(16, 32)[:, 16:] == y1)


@pytest.mark.gpu
def test_pcm_hand_vector_on_the_gpu(hot):
    rec, *_ = _pcm_records()
    yuv, _ = hot.recon_host(_flat_params(2, 2), rec, 1)
    _check_pcm_picture(yuv)


@pytest.mark.gpu
@pytest.mark.parametrize("profile,slices,pcm,scaling", CASES)
def test_kernels_match_the_oracle_on_generated_streams(hot, profile, slices, pcm, scaling):
    """every layout is asked for (the `hot` fixture); slices / scaling batches run on the one-picture kernel ("rows", or in bands: "wide") whatever was asked"""
    W, H, F = 11, 7, 5
    stream, packed, _ = gen.make_stream_ex(W, H, F, seed=77 + slices + pcm, profile=profile, slices=slices, pcm_permille=pcm,
                                           scaling=scaling, qp_range=(10, 45))
    with Stream(stream, spec=True) as s:
        p = s.params(0)
        recs = np.stack([s.packed(k)[1].reshape(W * H, 800) for k in range(F)])
    assert np.array_equal(recs, packed)
    yuv, rgb = hot.recon_host(p, recs, F, want_rgb=True)
    ref_yuv, ref_rgb = loader.recon(p, recs, F, want_rgb=True)
    assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb)
    if p.flags & (SLICES | SCALING):   # a one-picture kernel: one workgroup per picture / in bands when asked for, else three waves per row
        assert hot.last_launch()[0] in ("rows", "wide", "pipe1")


@pytest.mark.gpu
def test_slices_change_the_pictures(hot):
    """the availability bits matter: the same records reconstructed with and without MVHP_PARAM_SLICES differ"""
    W, H = 9, 6
    stream, packed, _ = gen.make_stream_ex(W, H, 2, seed=5, profile="baseline", slices=7)
    with Stream(stream, spec=True) as s:
        p = s.params(0)
    a, _ = hot.recon_host(p, packed, 2)
    q = _flat_params(W, H, p.flags & ~SLICES)
    b, _ = hot.recon_host(q, packed, 2)
    assert not np.array_equal(a, b)
    assert np.array_equal(a, loader.recon(p, packed, 2)[0]) and np.array_equal(b, loader.recon(q, packed, 2)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("profile,slices,pcm,scaling", [("high", 4, 50, 3), ("baseline", 3, 100, 0)])
def test_engine_and_cli_on_spec_streams(tmp_path, profile, slices, pcm, scaling):
    """the whole product path (MINIVIDEO_SPEC=1): stream bytes -> engine -> files, equal to the oracle on the generator's records"""
    from minivideo_amd import Engine
    W, H, F = 8, 5, 9
    stream, packed, _ = gen.make_stream_ex(W, H, F, seed=31, profile=profile, slices=slices, pcm_permille=pcm, scaling=scaling)
    got = {}

    def sink(seq, idr, rc, err, pr, yuv, rgb):
        got[seq] = (rc, yuv.copy() if yuv is not None else None)
        return 1 if rc == 1 else 0

    with Stream(stream, spec=True) as s:
        p = s.params(0)
        eng = Engine(contexts=1)
        rc, st = eng.decode(s.h, list(range(F)), want_rgb=False, sink=sink)
        eng.close()
    assert rc == 1 and st["pictures_ok"] == F
    for k in range(F):
        assert np.array_equal(got[k][1], loader.recon(p, packed[k], 1)[0]), k
    path = tmp_path / "clip.264"
    stream.tofile(path)
    cli = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")
    r = subprocess.run([cli, "-i", str(path), "-f", "yuv420", "-n", str(F)], cwd=tmp_path, capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, MINIVIDEO_SPEC="1"))
    assert r.returncode == 0, r.stderr
    for k in range(F):
        assert np.array_equal(np.fromfile(tmp_path / f"clip_{k}.yuv", np.uint8), loader.recon(p, packed[k], 1)[0]), k
