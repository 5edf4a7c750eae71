"""CPU: the host front end must survive corrupted input (no crash, no hang, a clean return code) -- the reference
calls exit() or reads out of bounds on several of these."""
import numpy as np
import pytest

from minivideo_amd import gen
from tests.util import Stream


@pytest.mark.parametrize("profile", ["baseline", "high", "high_cavlc"])
def test_bit_flips_never_crash(profile):
    stream, _ = gen.make_stream(6, 5, 2, seed=11, profile=profile)
    base = stream.copy()
    rng = np.random.default_rng(12345)
    outcomes = {1: 0, 0: 0, -1: 0}
    for trial in range(150):
        data = base.copy()
        n_flips = int(rng.integers(1, 6))
        for _ in range(n_flips):
            pos = int(rng.integers(0, data.size - 64))
            data[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        with Stream(data) as s:
            if not s.ok:
                continue
            for k in range(s.idr_count):
                rc, packed = s.packed(k)
                assert rc in (1, 0, -1)
                outcomes[rc] += 1
                if rc == 1:
                    rec = packed.reshape(-1, 800)
                    assert rec[:, 0].max() <= 2            # mb_kind
                    assert rec[:, 1].max() <= 51           # QP
                    assert rec[:, 12:28].max() <= 8        # prediction modes
    assert outcomes[1] > 0 and outcomes[0] + outcomes[-1] > 0


def test_truncations_never_crash():
    stream, _ = gen.make_stream(8, 6, 1, seed=3, profile="high")
    b = stream.tobytes()
    start = b.index(b"\x00\x00\x00\x01\x65") + 5
    for cut in range(start, len(b) - 64, 97):
        data = np.frombuffer(b[:cut] + bytes(64), np.uint8)
        with Stream(data) as s:
            if s.ok and s.idr_count:
                rc, _ = s.packed(0)
                assert rc in (1, 0, -1)


def test_garbage_parameter_sets():
    rng = np.random.default_rng(7)
    for _ in range(100):
        body = rng.integers(0, 256, size=24, dtype=np.uint8).tobytes()
        data = b"\x00\x00\x00\x01\x67" + body + b"\x00\x00\x00\x01\x68" + body[:8] + b"\x00\x00\x00\x01\x65" + body + bytes(64)
        with Stream(np.frombuffer(data, np.uint8)) as s:
            if s.ok and s.idr_count:
                rc, _ = s.packed(0)
                assert rc in (1, 0, -1)


@pytest.mark.parametrize("profile,slices,pcm,scaling", [("baseline", 3, 120, 0), ("high", 3, 60, 3), ("high_cavlc", 2, 150, 2)])
def test_spec_mode_bit_flips_never_crash(profile, slices, pcm, scaling):
    """round 3: the MVHP_STREAM_SPEC parsers (scaling lists, slice grouping, I_PCM incl. the CABAC restart) on corrupted input;
    the sanitizer version of this is tools/fuzz_frontend.cpp ... spec (20 000 mutated streams clean under ASan + UBSan)"""
    from tests.compact import decode_compact
    stream, _, _ = gen.make_stream_ex(6, 5, 2, seed=21, profile=profile, slices=slices, pcm_permille=pcm, scaling=scaling)
    rng = np.random.default_rng(99)
    seen = {1: 0, 0: 0, -1: 0}
    for trial in range(120):
        data = stream.copy()
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, data.size - 64))
            data[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        with Stream(data, spec=True) as s:
            if not s.ok:
                continue
            for k in range(s.idr_count):
                rc, packed = s.packed(k)
                assert rc in (1, 0, -1)
                seen[rc] += 1
                if rc == 1:
                    rec = packed.reshape(-1, 800)
                    assert rec[:, 0].max() <= 3 and rec[:, 6].max() <= 15      # mb_kind incl. I_PCM, unavail bits
                    not_pcm = rec[:, 0] != 3
                    assert rec[not_pcm, 12:28].max(initial=0) <= 8
                rc2, used, buf = decode_compact(s, k)
                assert rc2 in (1, 0, -1) and (buf is None or used <= buf.size)
    assert seen[1] > 0 and seen[0] + seen[-1] > 0
