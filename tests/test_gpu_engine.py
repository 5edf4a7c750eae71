"""GPU: the decode engine (mvhp_engine_*, the pipeline behind minivideo_decode) against the oracle --
stream bytes -> entropy threads -> H2D -> batched kernels -> D2H -> sink, in order, over 1-3 contexts on the one
device of the test box (the multi-device code path: one uploader / launcher / downloader per context pulling whole
batches from one queue), with an injected device failure (re-queue to another context), with `wanted` capping the
entropy work, and with batches large enough to reach the four- and eight-picture kernels."""
import os
import subprocess

import numpy as np
import pytest

from minivideo_amd import Engine, gen
from minivideo_amd.hotpath import StreamParams
from oracle import loader
from tests.util import Stream

pytestmark = pytest.mark.gpu


def _cus():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")


def _decode_all(eng, stream, n, want_rgb=True, wanted=None, order=None):
    got = {}
    seqs = []

    def sink(seq, idr, rc, err, p, yuv, rgb):
        seqs.append(seq)
        if rc != 1:
            got[seq] = (idr, rc, err, None, None)
            return 0
        got[seq] = (idr, rc, err, yuv.copy(), rgb.copy() if rgb is not None else None)
        return 1

    with Stream(stream) as s:
        assert s.ok and s.idr_count == n
        rc, st = eng.decode(s.h, list(range(n)) if order is None else order, wanted=wanted, want_rgb=want_rgb, sink=sink)
    assert seqs == list(range(len(seqs)))   # the sink is called in order
    return rc, st, got


@pytest.mark.parametrize("contexts", [1, 2, 3])
@pytest.mark.parametrize("profile", ["baseline", "high"])
def test_engine_matches_oracle(contexts, profile):
    W, H, F = 20, 12, 70
    stream, packed = gen.make_stream(W, H, F, seed=91 + contexts, profile=profile)
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=contexts, chunk_pictures=5, batch_pictures=16)
    rc, st, got = _decode_all(eng, stream, F)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == F and st["pictures_failed"] == 0 and st["contexts"] == contexts
    assert st["batches"] >= F // 16 and st["pictures_issued"] == F
    for k in range(F):
        ref_yuv, ref_rgb = loader.recon(p, packed[k], 1, want_rgb=True)
        assert got[k][0] == k
        assert np.array_equal(got[k][3], ref_yuv), k
        assert np.array_equal(got[k][4], ref_rgb), k


def test_engine_requeues_a_failed_batch():
    W, H, F = 12, 8, 64
    stream, packed = gen.make_stream(W, H, F, seed=77, profile="high")
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=3, chunk_pictures=4, batch_pictures=8, fail_context=0)
    rc, st, got = _decode_all(eng, stream, F, want_rgb=False)
    eng.close()
    assert rc == 1 and st["batches_requeued"] == 1 and st["pictures_ok"] == F and st["pictures_issued"] == F + 8
    for k in range(F):
        assert np.array_equal(got[k][3], loader.recon(p, packed[k], 1)[0]), k


def test_engine_single_context_failure_is_reported():
    W, H, F = 12, 8, 24
    stream, _ = gen.make_stream(W, H, F, seed=78, profile="baseline", want_packed=False)
    eng = Engine(contexts=1, chunk_pictures=4, batch_pictures=8, fail_context=0)
    rc, st, got = _decode_all(eng, stream, F, want_rgb=False)
    eng.close()
    assert rc == 1 and st["pictures_failed"] == 8 and st["pictures_ok"] == F - 8
    assert all(got[k][1] != 1 and "injected" in got[k][2] for k in range(8))


def test_engine_wanted_caps_entropy_work():
    W, H, F = 12, 8, 50
    stream, packed = gen.make_stream(W, H, F, seed=79, profile="main")
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=2)
    rc, st, got = _decode_all(eng, stream, F, wanted=3)
    assert rc == 1 and st["pictures_issued"] == 3 and st["pictures_ok"] == 3 and len(got) == 3
    for k in range(3):
        assert np.array_equal(got[k][3], loader.recon(p, packed[k], 1)[0])
    # the engine is reusable: a second call, arbitrary order
    order = [7, 3, 3, 49, 0]
    rc, st, got = _decode_all(eng, stream, F, order=order, want_rgb=False)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == 5
    for i, k in enumerate(order):
        assert got[i][0] == k and np.array_equal(got[i][3], loader.recon(p, packed[k], 1)[0])


@pytest.mark.parametrize("batch,forced,layout", [(1024, None, "pipe1"), (2048, "wide", "wide"), (2048, "quad_wide", "quad_wide"),
                                                 (1024, "quad", "quad"), (2048, "oct", "oct")])
def test_engine_reaches_the_batch_kernels(batch, forced, layout, monkeypatch):
    """what pick_layout (hotpath_abi.hip) makes of the engine's batches of SMALL pictures (6 rows: the ramp's batches of 16 ... 512
    are at most 3072 row-waves -> one picture per wavefront, three waves per row), and every other form forced through the
    environment (the one-workgroup-per-group kernels are the automatic choice from ~860 pictures per launch on)"""
    W, H, D = 12, 6, 32
    F = 2080
    if forced:
        monkeypatch.setenv("MINIVIDEO_LAYOUT", forced)
    stream, packed = gen.make_stream(W, H, D, seed=80, profile="baseline")
    # the same D pictures over and over: SPS/PPS once, then the IDR NAL units repeated
    starts = [i for i in range(len(stream) - 4) if stream[i] == 0 and stream[i + 1] == 0 and stream[i + 2] == 0 and stream[i + 3] == 1]
    first_idr = next(i for i in starts if stream[i + 4] == 0x65)
    body = stream[first_idr:len(stream) - 64]
    big = np.concatenate([stream[:first_idr]] + [body] * (F // D) + [stream[len(stream) - 64:]])
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=1, batch_pictures=batch)
    hashes = {}

    def sink(seq, idr, rc, err, pr, yuv, rgb):
        if rc == 1 and (seq % 97 == 0 or seq >= F - 2):
            hashes[seq] = (yuv.copy(), rgb.copy())
        return 1 if rc == 1 else 0

    with Stream(big) as s:
        assert s.idr_count == F
        rc, st = eng.decode(s.h, list(range(F)), want_rgb=True, sink=sink)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == F
    by_layout = dict(zip(["auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"], list(st["launches_by_layout"]) + list(st["launches_wide"])))
    if forced or _cus() == 256:
        assert by_layout[layout] >= 1, by_layout
    for seq, (yuv, rgb) in hashes.items():
        ref_yuv, ref_rgb = loader.recon(p, packed[seq % D], 1, want_rgb=True)
        assert np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb), seq


def _cli(tmp_path, stream, args, env=None):
    path = tmp_path / "clip.264"
    stream.tofile(path)
    e = dict(os.environ, MINIVIDEO_STATS="1", **(env or {}))
    r = subprocess.run([CLI, "-i", str(path), *args], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0, r.stderr
    assert "decode did not succeed" not in r.stderr, r.stderr
    return r


def test_cli_one_thumbnail_decodes_one_picture(tmp_path):
    """picture_number = 1 on a stream of many IDRs must not entropy-decode the stream."""
    W, H, F = 20, 12, 40
    stream, packed = gen.make_stream(W, H, F, seed=81, profile="baseline")
    r = _cli(tmp_path, stream, ["-f", "yuv420", "-n", "1"])
    assert "decode: 1 pictures entropy-decoded, 1 written" in r.stderr, r.stderr
    got = np.fromfile(tmp_path / "clip.yuv", np.uint8)
    assert np.array_equal(got, loader.recon(StreamParams(W, H, 0, 0, 0), packed[0], 1)[0])


@pytest.mark.parametrize("fake", [2, 3])
def test_cli_several_contexts(tmp_path, fake):
    """The multi-device deal-out through the public API: N contexts on the one device (MINIVIDEO_FAKE_GPUS)."""
    W, H, F = 12, 8, 66
    stream, packed = gen.make_stream(W, H, F, seed=82 + fake, profile="high")
    r = _cli(tmp_path, stream, ["-f", "yuv420", "-n", str(F)], env={"MINIVIDEO_FAKE_GPUS": str(fake), "MINIVIDEO_BATCH": "8"})
    assert f"{fake} contexts" in r.stderr and f"{F} written" in r.stderr, r.stderr
    p = StreamParams(W, H, 0, 0, 0)
    for k in range(F):
        got = np.fromfile(tmp_path / f"clip_{k}.yuv", np.uint8)
        assert np.array_equal(got, loader.recon(p, packed[k], 1)[0]), k


def test_cli_requeue_through_the_api(tmp_path):
    W, H, F = 12, 8, 40
    stream, packed = gen.make_stream(W, H, F, seed=85, profile="baseline")
    r = _cli(tmp_path, stream, ["-f", "bmp", "-n", str(F)],
             env={"MINIVIDEO_FAKE_GPUS": "2", "MINIVIDEO_BATCH": "8", "MINIVIDEO_TEST_FAIL_CONTEXT": "1"})
    assert f"{F} written" in r.stderr and f"{F + 8} pictures entropy-decoded" in r.stderr, r.stderr
    assert all(os.path.exists(tmp_path / f"clip_{k}.bmp") for k in range(F))


@pytest.mark.parametrize("profile,threshold", [("baseline", None), ("high", None), ("high", "12"), ("main", "0")])
def test_expand_kernel_matches_the_packed_records(profile, threshold, monkeypatch):
    """mvhp_expand_compact_dev: compact pictures (what crosses PCIe) -> packed records, byte for byte what
    mvhp_stream_decode_packed writes -- also with macroblocks sent as dense coefficient areas (threshold lowered through
    the test hook; 0 = every macroblock that has a level)."""
    import torch
    from minivideo_amd import HotPath
    from tests.compact import COMPACT_MB_BYTES_MAX, COMPACT_SLACK_BYTES, decode_compact
    if threshold is not None:
        monkeypatch.setenv("MINIVIDEO_TEST_COMPACT_MAX", threshold)
    W, H, F = 13, 7, 5
    stream, packed = gen.make_stream(W, H, F, seed=88, profile=profile)
    stride = (W * H * COMPACT_MB_BYTES_MAX + COMPACT_SLACK_BYTES + 15) & ~15
    host = np.zeros((F, stride), np.uint8)
    with Stream(stream) as s:
        p = s.params(0)
        for k in range(F):
            rc, used, buf = decode_compact(s, k)
            assert rc == 1
            host[k, :used] = buf[:used]
    d_compact = torch.from_numpy(host).cuda()
    d_packed = torch.full((F * W * H * 800,), 0xEE, dtype=torch.uint8, device="cuda")
    hot = HotPath(0)
    hot.expand_compact_dev(p, d_compact.data_ptr(), stride, F, d_packed.data_ptr())
    hot.sync_check()
    hot.close()
    got = d_packed.cpu().numpy().reshape(F, W * H, 800)
    assert np.array_equal(got, packed)


def test_engine_rgb_only_skips_the_plane_download():
    """MVHP_OUT_RGB_ONLY (what minivideo_decode asks for when it writes bmp / png / tga): the sink gets RGB and no planes,
    and only the RGB bytes cross the link"""
    W, H, F = 12, 8, 20
    stream, packed = gen.make_stream(W, H, F, seed=93, profile="high")
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=1)
    got = {}

    def sink(seq, idr, rc, err, pr, yuv, rgb):
        got[seq] = (rc, yuv is None, rgb.copy())
        return 1

    with Stream(stream) as s:
        rc, st = eng.decode(s.h, list(range(F)), want_rgb=3, sink=sink)
    eng.close()
    assert rc == 1 and st["pictures_ok"] == F and st["d2h_bytes"] == F * p.rgb_bytes
    for k in range(F):
        assert got[k][0] == 1 and got[k][1]
        assert np.array_equal(got[k][2], loader.recon(p, packed[k], 1, want_rgb=True)[1]), k


def test_engine_kept_pictures_are_valid_until_released():
    """round 3: a sink that answers 2 keeps the picture; a second thread compares it with the oracle LATER and gives it back
    (mvhp_engine_release_picture).  Small chunks and batches so that kept pictures pin output chunks the downloader wants."""
    import queue
    import threading
    W, H, F = 20, 12, 90
    stream, packed = gen.make_stream(W, H, F, seed=97, profile="high")
    p = StreamParams(W, H, 0, 0, 0)
    eng = Engine(contexts=2, chunk_pictures=4, batch_pictures=12)
    q = queue.Queue()
    bad, released = [], []

    def checker():
        while True:
            item = q.get()
            if item is None:
                return
            seq, yuv, rgb = item
            ref_yuv, ref_rgb = loader.recon(p, packed[seq], 1, want_rgb=True)     # (takes longer than the pipeline needs per picture)
            if not (np.array_equal(yuv, ref_yuv) and np.array_equal(rgb, ref_rgb)):
                bad.append(seq)
            released.append(seq)
            eng.release_picture(seq)

    th = threading.Thread(target=checker)
    th.start()

    def sink(seq, idr, rc, err, prm, yuv, rgb):
        if rc != 1:
            bad.append(seq)
            return 0
        if seq % 4 == 3:
            return 1
        q.put((seq, yuv, rgb))     # views, not copies
        return 2

    with Stream(stream) as s:
        rc, st = eng.decode(s.h, list(range(F)), want_rgb=True, sink=sink)
        n_released_at_return = len(released)
    q.put(None)
    th.join()
    eng.close()
    kept = F - F // 4
    assert rc == 1 and st["pictures_ok"] == F and not bad
    assert n_released_at_return == kept     # the call waited for the last kept picture
