"""GPU: the public API end to end (open -> parse -> decode -> files in the CWD) through the CLI,
against files built from the oracle's planes.  BMP/TGA byte layouts are restated here from the
stb_image_write v1.01 formats the reference emits (export.c:447-606)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from minivideo_amd import gen
from oracle import loader
from minivideo_amd.hotpath import StreamParams

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")


def _run(tmp_path, stream, name, *args, env=None):
    path = tmp_path / name
    stream.tofile(path)
    r = subprocess.run([CLI, "-i", str(path), *args], cwd=tmp_path, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stderr
    assert "decode did not succeed" not in r.stderr, r.stderr
    return r


def _expected(W, H, packed_frames, want_rgb=False):
    p = StreamParams(W, H, 0, 0, 0)
    return [loader.recon(p, f, 1, want_rgb=want_rgb) for f in packed_frames]


def _bmp(rgb, w, h):
    pad = (-w * 3) & 3
    out = b"BM" + struct.pack("<IHHI", 14 + 40 + (w * 3 + pad) * h, 0, 0, 54)
    out += struct.pack("<IIIHH", 40, w, h, 1, 24) + struct.pack("<6I", 0, 0, 0, 0, 0, 0)
    img = rgb.reshape(h, w, 3)[::-1, :, ::-1]
    rows = [img[y].tobytes() + bytes(pad) for y in range(h)]
    return out + b"".join(rows)


def _tga(rgb, w, h):
    out = bytearray(struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, w, h, 24, 0))
    img = rgb.reshape(h, w, 3)
    for j in range(h - 1, -1, -1):
        row = [bytes(img[j, i]) for i in range(w)]
        i = 0
        while i < w:
            ln, diff = 1, True
            if i < w - 1:
                ln = 2
                diff = row[i] != row[i + 1]
                if diff:
                    prev = i
                    k = i + 2
                    while k < w and ln < 128:
                        if row[prev] != row[k]:
                            prev += 1
                            ln += 1
                        else:
                            ln -= 1
                            break
                        k += 1
                else:
                    k = i + 2
                    while k < w and ln < 128:
                        if row[i] == row[k]:
                            ln += 1
                        else:
                            break
                        k += 1
            if diff:
                out.append(ln - 1)
                for k in range(ln):
                    out += row[i + k][::-1]
            else:
                out.append((ln - 129) & 255)
                out += row[i][::-1]
            i += ln
    return bytes(out)


def _png_pixels(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + body) & 0xffffffff)
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 3 + 1)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(-1), w, h


@pytest.mark.parametrize("profile", ["baseline", "high"])
def test_cli_yuv420_multi(tmp_path, profile):
    W, H, F = 20, 12, 5
    stream, packed = gen.make_stream(W, H, F, seed=31, profile=profile)
    _run(tmp_path, stream, "clip.264", "-f", "yuv420", "-n", str(F))
    exp = _expected(W, H, packed)
    for k in range(F):
        got = np.fromfile(tmp_path / f"clip_{k}.yuv", np.uint8)
        assert np.array_equal(got, exp[k][0]), k


@pytest.mark.parametrize("writers", ["0", "1", "6"])
@pytest.mark.parametrize("fmt", ["yuv420", "bmp"])
def test_cli_file_writers(tmp_path, writers, fmt):
    """round 3: minivideo_decode's sink keeps the picture and a pool of threads writes the files (MINIVIDEO_WRITERS, 0 = on
    the calling thread as before): same names, same bytes, every picture once, whatever the pool size"""
    W, H, F = 22, 14, 60
    stream, packed = gen.make_stream(W, H, F, seed=61, profile="high")
    r = _run(tmp_path, stream, "w.264", "-f", fmt, "-n", str(F), env={"MINIVIDEO_WRITERS": writers, "MINIVIDEO_STATS": "1"})
    assert f"{writers} file writers" in r.stderr and f"{F} written" in r.stderr, r.stderr
    exp = _expected(W, H, packed, want_rgb=(fmt == "bmp"))
    for k in range(F):
        if fmt == "bmp":
            assert (tmp_path / f"w_{k}.bmp").read_bytes() == _bmp(exp[k][1], W * 16, H * 16), k
        else:
            assert np.array_equal(np.fromfile(tmp_path / f"w_{k}.yuv", np.uint8), exp[k][0]), k
    assert not (tmp_path / f"w_{F}.{fmt[:3]}").exists()


def test_cli_unwritable_directory_fails(tmp_path):
    """every write fails (read-only working directory): the pool reports the files it tried, decoding stops once a failure is
    known (a full disk does not get better) and the call fails (minivideo.h: the contract of minivideo_decode)"""
    import stat
    stream, _ = gen.make_stream(6, 4, 8, seed=62, profile="baseline")
    (tmp_path / "in").mkdir()
    src = tmp_path / "in" / "x.264"
    stream.tofile(src)
    ro = tmp_path / "ro"
    ro.mkdir()
    os.chmod(ro, stat.S_IRUSR | stat.S_IXUSR)
    try:
        if os.access(ro, os.W_OK):
            pytest.skip("running as a user that ignores directory permissions")
        r = subprocess.run([CLI, "-i", str(src), "-f", "yuv420", "-n", "8"], cwd=ro, capture_output=True, text=True, timeout=120)
        assert 1 <= r.stderr.count("Unable to write") <= 8, r.stderr
        assert "decode did not succeed" in r.stderr, (r.stdout, r.stderr)
    finally:
        os.chmod(ro, stat.S_IRWXU)


def test_cli_mp4_equals_es(tmp_path):
    from tests.mp4mux import mux
    W, H, F = 12, 9, 4
    stream, packed = gen.make_stream(W, H, F, seed=37, profile="high")
    data = np.frombuffer(mux(stream, W * 16, H * 16, extra_non_sync=True, samples_per_chunk=3), np.uint8)
    _run(tmp_path, data, "clip.mp4", "-f", "yuv420", "-n", str(F))
    exp = _expected(W, H, packed)
    for k in range(F):
        assert np.array_equal(np.fromfile(tmp_path / f"clip_{k}.yuv", np.uint8), exp[k][0]), k


def test_cli_single_picture_name_and_full_hd(tmp_path):
    stream, packed = gen.make_stream(120, 68, 2, seed=32, profile="baseline")
    _run(tmp_path, stream, "movie.h264", "-f", "yuv420")       # -n 1: no _k suffix (export.c:630-642)
    got = np.fromfile(tmp_path / "movie.yuv", np.uint8)
    assert got.size == 3133440
    assert np.array_equal(got, _expected(120, 68, packed[:1])[0][0])
    assert not (tmp_path / "movie_0.yuv").exists()


def test_cli_rgb_writers(tmp_path):
    W, H = 7, 5                                              # width 112: row padding 0; odd MB counts
    stream, packed = gen.make_stream(W, H, 1, seed=33, profile="high")
    yuv, rgb = _expected(W, H, packed, want_rgb=True)[0]
    _run(tmp_path, stream, "p.264", "-f", "bmp")
    assert (tmp_path / "p.bmp").read_bytes() == _bmp(rgb, W * 16, H * 16)
    _run(tmp_path, stream, "p.264", "-f", "tga")
    assert (tmp_path / "p.tga").read_bytes() == _tga(rgb, W * 16, H * 16)
    _run(tmp_path, stream, "p.264", "-f", "png")
    px, w, h = _png_pixels((tmp_path / "p.png").read_bytes())
    assert (w, h) == (W * 16, H * 16) and np.array_equal(px, rgb)
    _run(tmp_path, stream, "p.264")                           # default jpg falls back to png (export.c:652-658)
    px2, _, _ = _png_pixels((tmp_path / "p.png").read_bytes())
    assert np.array_equal(px2, rgb)


def test_cli_yuv444_keeps_reference_hole(tmp_path):
    W, H = 4, 3
    stream, packed = gen.make_stream(W, H, 1, seed=34, profile="baseline")
    yuv = _expected(W, H, packed)[0][0]
    _run(tmp_path, stream, "q.264", "-f", "yuv444")
    got = np.fromfile(tmp_path / "q.yuv", np.uint8)
    w, h = W * 16, H * 16
    assert got.size == 3 * w * h
    assert np.array_equal(got[: w * h], yuv[: w * h])
    cb = yuv[w * h: w * h + w * h // 4].reshape(h // 2, w // 2)
    up = got[w * h: 2 * w * h].reshape(h, w)
    assert np.array_equal(up[0::2, 0::2], cb) and np.array_equal(up[0::2, 1::2], cb) and np.array_equal(up[1::2, 0::2], cb)
    assert np.all(up[1::2, 1::2] == 0)                       # export.c:267-268


def test_cli_skips_broken_picture(tmp_path):
    W, H, F = 6, 4, 4
    stream, packed = gen.make_stream(W, H, F, seed=35, profile="baseline")
    b = bytearray(stream.tobytes())
    # corrupt the 2nd IDR's slice header: slice_type field -> P (fails the IDR check), decoding continues
    idx = [i for i in range(len(b) - 5) if b[i:i + 5] == b"\x00\x00\x00\x01\x65"]
    b[idx[1] + 5] = 0xA0
    _run(tmp_path, np.frombuffer(bytes(b), np.uint8), "s.264", "-f", "yuv420", "-n", "3")
    exp = _expected(W, H, packed)
    for k, f in enumerate([0, 2, 3]):                         # picture 1 skipped; files numbered by successes
        assert np.array_equal(np.fromfile(tmp_path / f"s_{k}.yuv", np.uint8), exp[f][0])


def test_stream_to_gpu_matches_oracle_high_4k(hot):
    from tests.util import Stream
    stream, packed = gen.make_stream(240, 135, 1, seed=36, profile="high")
    with Stream(stream) as s:
        p = s.params(0)
        rc, got = s.packed(0)
        assert rc == 1
    yuv, rgb = hot.recon_host(p, got, 1, want_rgb=True)
    ryuv, rrgb = loader.recon(p, got, 1, want_rgb=True)
    assert np.array_equal(yuv, ryuv) and np.array_equal(rgb, rrgb)
    assert yuv.size == 12441600


def _idr_sample_sizes(stream_bytes):
    """Sample table of the ES parser (esparser.c:40-143): indexed NALs are 00 00 00 01 + {0x65,0x67,0x68}; a sample runs
    from its NAL header byte to the next indexed NAL header (or EOF); the scan stops 32 bytes before EOF."""
    b = stream_bytes
    pos = [i + 4 for i in range(len(b) - 32) if b[i:i + 4] == b"\x00\x00\x00\x01" and b[i + 4] in (0x65, 0x67, 0x68)]
    sizes = []
    for k, p in enumerate(pos):
        end = pos[k + 1] if k + 1 < len(pos) else len(b)
        if b[p] == 0x65:
            sizes.append(end - p)
    return sizes


@pytest.mark.parametrize("mode", ["ordered", "distributed"])
def test_cli_idr_selection_modes(tmp_path, mode):
    """filter.c:94-211: keep IDR pictures larger than mean/1.66, then the first N (ordered) or every (T/(N-1))-th."""
    W, H, F, N = 6, 4, 10, 3
    # alternate dense and light pictures so that the size filter has something to drop
    parts, packed = [], []
    for k in range(F):
        st, pk = gen.make_stream(W, H, 1, seed=50 + k, profile="baseline", dense=(k % 3 != 1))
        b = st.tobytes()
        if k:   # one SPS/PPS at the start of the file only
            b = b[b.index(b"\x00\x00\x00\x01\x65"):]
        parts.append(b.rstrip(b"\x00") if k < F - 1 else b)
        packed.append(pk[0])
    data = b"".join(parts)
    sizes = _idr_sample_sizes(data)
    assert len(sizes) == F
    thr = int((sum(sizes) / F) / 1.66)
    cand = [i for i in range(F) if sizes[i] > thr]
    assert 0 < len(cand) < F
    n = min(N, len(cand))
    jump = len(cand) // (n - 1)
    want = [cand[min(i if mode == "ordered" else i * jump, len(cand) - 1)] for i in range(n)]
    _run(tmp_path, np.frombuffer(data, np.uint8), "sel.264", "-f", "yuv420", "-n", str(N), "-e", mode)
    exp = _expected(W, H, packed)
    for k, f in enumerate(want):
        assert np.array_equal(np.fromfile(tmp_path / f"sel_{k}.yuv", np.uint8), exp[f][0]), (k, f)
