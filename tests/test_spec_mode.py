"""SURVEY 8f row f4, opt-in (MVHP_STREAM_SPEC / MINIVIDEO_SPEC=1): behaviour outside the reference's acceptance
envelope, by the standard instead of by the reference's quirks.  NOT part of the parity contract -- the expected values
here are hand-derived from the standard's text, the reference gives different answers (or none) on these inputs:

* Annex B byte streams with three-byte start codes, nal_ref_idc != 3 and no zero padding behind the last NAL unit
  (esparser.c:65-82 indexes none of it);
* Intra16x16 luma DC at QP'Y = 36: the standard's `qP >= 36` branch (8.5.10) gives Y = 136 for the Appendix A
  macroblock where the reference's `qP > 36` (h264_transform.c:797-808) gives 0.
  Derivation (one DC level +3, everything else zero): f = 3 in all 16 positions; LevelScale4x4(36 % 6 = 0, 0, 0) =
  16 * 10 = 160; standard: dcY = (3 * 160) << (36/6 - 6) = 480; the block's only coefficient d00 = 480 -> every
  residual = (480 + 32) >> 6 = 8 -> 128 + 8 = 136.  (QP 35: (3*176 + 2^0) >> 1 = 264 -> (264+32)>>6 = 4 ... the
  reference-mode values 135 / 0 / 136 for QP 35 / 36 / 37 stay as SURVEY Appendix A recorded them.)
* pictures of several slices, scaling matrices and I_PCM: tests/test_spec_f4.py (round 3)."""
import os

import numpy as np
import pytest

from minivideo_amd.hotpath import StreamParams
from oracle import loader
from tests.kat import kat_packed
from tests.util import Stream

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _kat_three_byte_start_codes():
    raw = np.fromfile(os.path.join(GOLDEN, "kat_cavlc_2mb.264"), np.uint8).tobytes()
    raw = raw.rstrip(b"\x00")                                    # no padding behind the last NAL unit
    out = raw.replace(b"\x00\x00\x00\x01\x67", b"\x00\x00\x01\x27")  # SPS, nal_ref_idc 1, three-byte start code
    out = out.replace(b"\x00\x00\x00\x01\x68", b"\x00\x00\x01\x28")  # PPS
    out = out.replace(b"\x00\x00\x00\x01\x65", b"\x00\x00\x01\x25")  # IDR slice, nal_ref_idc 1
    return np.frombuffer(out, np.uint8)


def test_reference_mode_does_not_index_three_byte_start_codes():
    with Stream(_kat_three_byte_start_codes()) as s:
        assert not s.ok or s.idr_count == 0


def test_spec_mode_indexes_and_parses_the_same_records():
    with Stream(np.fromfile(os.path.join(GOLDEN, "kat_cavlc_2mb.264"), np.uint8)) as ref:
        rc0, want = ref.packed(0)
    with Stream(_kat_three_byte_start_codes(), spec=True) as s:
        assert s.ok and s.idr_count == 1
        p = s.params(0)
        assert (p.width_mbs, p.height_mbs) == (2, 1) and (p.flags & 2)   # MVHP_PARAM_SPEC_LUMA_DC rides along
        rc, got = s.packed(0)
    assert rc0 == 1 and rc == 1 and np.array_equal(got, want)


def test_spec_mode_reads_four_byte_streams_too():
    data = np.fromfile(os.path.join(GOLDEN, "kat_cabac_2mb.264"), np.uint8)
    with Stream(data) as a, Stream(data, spec=True) as b:
        assert a.idr_count == b.idr_count == 1
        assert np.array_equal(a.packed(0)[1], b.packed(0)[1])


def test_spec_mode_groups_slices_and_refuses_overlapping_ones():
    """a second slice NAL with first_mb_in_slice > 0 belongs to the previous picture (round 3: decoded, tests/test_spec_f4.py);
    here it claims macroblock 1, which the first slice has already decoded: the picture fails, it is not mis-decoded"""
    raw = np.fromfile(os.path.join(GOLDEN, "kat_cavlc_2mb.264"), np.uint8).tobytes().rstrip(b"\x00")
    # first_mb_in_slice = 1 ('010'), slice_type 7 ('0001000'), pps 0 ('1'), then arbitrary payload
    second = b"\x00\x00\x00\x01\x65" + bytes([0b01000010, 0b00110000, 0x80])
    with Stream(np.frombuffer(raw + second + bytes(64), np.uint8), spec=True) as s:
        assert s.ok and s.idr_count == 1
        p = s.params(0)
        assert p is not None and (p.flags & 4)                      # MVHP_PARAM_SLICES
        buf = np.zeros(2 * 800, np.uint8)
        assert s.L.mvhp_stream_decode_packed(s.h, 0, buf.ctypes.data, buf.size) != 1
        assert "does not start where the previous one ended" in s.error()
        assert not buf.any()
    # the reference's reading of the same bytes (default mode): the second NAL is a picture of its own, and broken
    with Stream(np.frombuffer(raw + second + bytes(64), np.uint8)) as s:
        assert s.idr_count == 2 and s.packed(0)[0] == 1 and s.packed(1)[0] != 1


def test_oracle_luma_dc_rule_at_qp36():
    p, rec = kat_packed(36)
    assert loader.recon(p, rec, 1)[0][0] == 0                      # reference: the `qP > 36` defect (Appendix A)
    ps = StreamParams(p.width_mbs, p.height_mbs, 0, 0, 2)          # MVHP_PARAM_SPEC_LUMA_DC
    yuv, _ = loader.recon(ps, rec, 1)
    assert np.all(yuv[:512] == 136) and np.all(yuv[512:] == 128)   # hand-derived above
    for qp, y in ((35, 135), (37, 136)):                           # every other QP: both rules agree
        p, rec = kat_packed(qp)
        assert loader.recon(StreamParams(2, 1, 0, 0, 2), rec, 1)[0][0] == y == loader.recon(p, rec, 1)[0][0]


@pytest.mark.gpu
def test_gpu_luma_dc_rule_at_qp36(hot):
    for qp in (30, 35, 36, 37, 42):
        _, rec = kat_packed(qp)
        for flags in (0, 2):
            p = StreamParams(2, 1, 0, 0, flags)
            ref_yuv, ref_rgb = loader.recon(p, rec, 1, want_rgb=True)
            yuv, rgb = hot.recon_host(p, rec, 1, want_rgb=True)
            assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb), (qp, flags)
    p = StreamParams(2, 1, 0, 0, 2)
    assert hot.recon_host(p, kat_packed(36)[1], 1)[0][0] == 136


@pytest.mark.gpu
def test_gpu_spec_flag_random_pictures(hot):
    """the flag only changes Intra16x16 macroblocks at QP'Y = 36: random High pictures over QP 30..40, both settings"""
    from minivideo_amd.synth import synth_packed
    p, rec = synth_packed(9, 5, 3, seed=404, profile="high", density="dense")
    rec = rec.copy()
    rec[:, ::3, 1] = 36                                            # plenty of QP 36 macroblocks of every kind
    for flags in (0, 2):
        ps = StreamParams(p.width_mbs, p.height_mbs, 0, 0, flags | (p.flags & 1))
        ref_yuv, ref_rgb = loader.recon(ps, rec, 3, want_rgb=True)
        yuv, rgb = hot.recon_host(ps, rec, 3, want_rgb=True)
        assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb), flags
