"""CPU: the picture writers of minivideo_decode (export.cpp; export.c:618-767 in the reference).  Round 3 rewrote the PNG writer
for speed (CRC-32 eight bytes per step, Adler-32 sixteen bytes per step with SSE2, one pass over the rows): every chunk's CRC, the
zlib stream (its Adler-32 is checked by zlib.decompress), the stored-block framing and the pixels, for sizes around the 16-byte and
65535-byte boundaries; and a short write (a full device) must make a writer fail instead of leaving a truncated file behind."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = tmp_path_factory.mktemp("export") / "export_check"
    host = os.path.join(ROOT, "minivideo_amd", "csrc", "host")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-I" + host,
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "export_check.cpp"),
                           os.path.join(host, "export.cpp"), "-o", str(exe)])
    return str(exe)


def _pixels(w, h, seed):
    i = np.arange(w * h * 3, dtype=np.uint64)
    return (((i * 2654435761 + seed) & 0xffffffff) >> 13).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(1, 1), (5, 1), (5, 3), (16, 16), (21, 7), (112, 80), (1920, 1088), (7281, 3), (7282, 3), (21845, 1), (0, 0)])
def test_png_is_a_valid_png_of_the_picture(tool, tmp_path, w, h):
    out = tmp_path / "p.png"
    seed = w * 131 + h
    assert subprocess.run([tool, "png", str(w), str(h), str(seed), str(out)]).returncode == 0
    d = out.read_bytes()
    assert d[:8] == bytes([137, 80, 78, 71, 13, 10, 26, 10])
    pos, idat, types = 8, b"", []
    while pos < len(d):
        n = struct.unpack(">I", d[pos:pos + 4])[0]
        t, body = d[pos + 4:pos + 8], d[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + body) & 0xffffffff == struct.unpack(">I", d[pos + 8 + n:pos + 12 + n])[0], t
        types.append(t)
        if t == b"IHDR":
            assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 2, 0, 0, 0)
        if t == b"IDAT":
            idat += body
        pos += 12 + n
    assert types == [b"IHDR", b"IDAT", b"IEND"]
    raw = zlib.decompress(idat)                       # (raises on a wrong Adler-32 or broken block framing)
    assert len(raw) == (3 * w + 1) * h
    rows = np.frombuffer(raw, np.uint8).reshape(h, 3 * w + 1) if h else np.zeros((0, 1), np.uint8)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(-1), _pixels(w, h, seed))
    # stored blocks of 65535 bytes, the last one flagged: the framing the reference-era writer produced
    assert idat[:2] == b"\x78\x01"
    p, left = 2, len(raw)
    while True:
        final, ln, nln = idat[p], idat[p + 1] | (idat[p + 2] << 8), idat[p + 3] | (idat[p + 4] << 8)
        assert ln == min(left, 65535) and nln == (~ln & 0xffff) and final == (1 if left <= 65535 else 0)
        p += 5 + ln
        left -= ln
        if final:
            break
    assert left == 0 and p + 4 == len(idat)


@pytest.mark.parametrize("fmt", ["png", "bmp", "tga"])
def test_a_short_write_fails_the_writer(tool, fmt):
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full here")
    assert subprocess.run([tool, fmt, "64", "48", "1", "/dev/full"]).returncode == 1


@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (6, 2), (7, 5), (11, 4), (21, 7), (112, 80), (1920, 8)])
def test_bmp_rows_are_bottom_up_bgr(tool, tmp_path, w, h):
    """stbi_write_bmp's layout (export.c:716-724); the R/B swap runs five pixels per SSSE3 shuffle with a scalar tail"""
    out = tmp_path / "p.bmp"
    seed = w * 17 + h
    assert subprocess.run([tool, "bmp", str(w), str(h), str(seed), str(out)]).returncode == 0
    d = out.read_bytes()
    pad = (-w * 3) & 3
    assert d[:2] == b"BM" and struct.unpack("<I", d[2:6])[0] == 54 + (w * 3 + pad) * h == len(d)
    assert struct.unpack("<IIIHH", d[14:30]) == (40, w, h, 1, 24)
    rows = np.frombuffer(d[54:], np.uint8).reshape(h, w * 3 + pad)
    px = _pixels(w, h, seed).reshape(h, w, 3)
    assert np.array_equal(rows[::-1, :w * 3].reshape(h, w, 3)[:, :, ::-1], px)
    assert (rows[:, w * 3:] == 0).all()


def _flat_pixels(w, h, seed):
    out = np.zeros((w * h, 3), np.uint8)
    run = left = 0
    for k in range(w * h):
        if left == 0:
            run += 1
            left = 1 + (run * 37 + seed) % 200
        for c in range(3):
            out[k, c] = (((run * 2654435761) & 0xffffffff) + c * 97 + seed & 0xffffffff) >> 11 & 255
        left -= 1
    return out.reshape(-1)


@pytest.mark.parametrize("w,h,flat", [(1, 1, 0), (7, 5, 0), (130, 3, 0), (300, 4, 1), (1920, 2, 0), (257, 3, 1), (640, 5, 1)])
def test_tga_is_stb_rle(tool, tmp_path, w, h, flat):
    """stbi_write_tga with RLE (export.c:726-733): byte for byte against the Python restatement the GPU CLI tests use, on noise
    (raw packets, written through the vector R/B swap) and on runs of 1..200 equal pixels (run-length packets, 128-pixel splits)"""
    from tests.test_gpu_api import _tga
    out = tmp_path / "p.tga"
    seed = w * 5 + h
    assert subprocess.run([tool, "tga", str(w), str(h), str(seed), str(out)] + (["flat"] if flat else [])).returncode == 0
    px = _flat_pixels(w, h, seed) if flat else _pixels(w, h, seed)
    assert out.read_bytes() == _tga(px, w, h)
