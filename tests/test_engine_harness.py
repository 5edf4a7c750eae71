"""CPU: the decode engine's threading (csrc/host/decode_engine.cpp) built without HIP against a stub device table and
run under ThreadSanitizer (tools/engine_harness.cpp): in-order delivery through chunks / batches / several contexts,
`wanted` caps the entropy work (one thumbnail = one picture decoded), a failed batch is re-queued once to
another context, a broken picture arrives as a failure in its place; pictures a sink keeps and other threads
release; and the public API (minivideo_open / parse / decode) with its file-writer pool.  No reconstruction happens here --
the stub stamps outputs with a checksum of the records it was handed."""
import os
import subprocess

import numpy as np
import pytest

from minivideo_amd import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "minivideo_amd", "csrc", "host")
SRC = [os.path.join(ROOT, "tools", "engine_harness.cpp")] + [os.path.join(HOST, f) for f in (
    "decode_engine.cpp", "stream_abi.cpp", "h264_frontend.cpp", "h264_cabac.cpp", "mp4_demux.cpp", "api.cpp", "export.cpp")]


def _build(tmp_path, sanitize):
    exe = tmp_path / ("harness_" + (sanitize or "plain"))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + HOST]
    if sanitize:
        cmd.append("-fsanitize=" + sanitize)
    subprocess.check_call(cmd + SRC + ["-o", str(exe)])
    return exe


def _break_picture(stream, k):
    """Overwrite the slice payload of IDR picture k with bytes that are no legal slice data."""
    data = stream.copy()
    starts = [i for i in range(len(data) - 4) if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 0 and data[i + 3] == 1
              and data[i + 4] == 0x65]
    a = starts[k] + 8
    b = (starts[k + 1] if k + 1 < len(starts) else len(data) - 64)
    data[a:b] = 0xFF
    return data


@pytest.mark.parametrize("sanitize", ["thread", "address,undefined"])
def test_engine_harness(tmp_path, sanitize):
    stream, _ = gen.make_stream(6, 4, 23, seed=5, profile="baseline", dense=True, want_packed=False)
    stream.tofile(tmp_path / "a.264")
    s2, _ = gen.make_stream(5, 3, 11, seed=6, profile="main", dense=True, want_packed=False)
    _break_picture(s2, 4).tofile(tmp_path / "b.264")
    exe = _build(tmp_path, sanitize)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([str(exe), str(tmp_path / "a.264"), str(tmp_path / "b.264"), "4"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "HARNESS OK" in r.stdout
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
