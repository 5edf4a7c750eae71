"""CPU, this container only: the reference's own mini_thumbnailer/src/main.cpp compiles and links,
unchanged and in place, against include/minivideo.h + libminivideo.so (drop-in boundary, config 0
plumbing).  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = "/root/reference/mini_thumbnailer/src/main.cpp"


@pytest.mark.skipif(not os.path.exists(MAIN), reason="reference checkout not present")
def test_stock_mini_thumbnailer_builds_and_runs(tmp_path):
    exe = tmp_path / "mini_thumbnailer_stock"
    pkg = os.path.join(ROOT, "minivideo_amd")
    subprocess.check_call(["g++", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(MAIN), MAIN,
                           "-L" + pkg, "-lminivideo", "-Wl,-rpath," + pkg, "-o", str(exe)])
    kat = os.path.join(ROOT, "tests", "golden", "kat_cavlc_2mb.264")
    r = subprocess.run([str(exe), "-i", kat, "-f", "yuv420"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0                     # exit status reflects minivideo_close (main.cpp:285-298)
    assert "Working..." in r.stdout
    import ctypes as C
    from minivideo_amd import lib
    if lib().mvhp_device_count() == 0:
        # no GPU here: decode must refuse loudly (no CPU reconstruction path), and write nothing
        assert "no HIP device" in r.stderr
        assert not (tmp_path / "kat_cavlc_2mb.yuv").exists()


def _enums(path):
    """{enum name: {constant: value}} of a C header, read as text (implicit values count up from the previous one)"""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(path, errors="replace").read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"typedef\s+enum\s+(\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        vals, nxt = {}, 0
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            name, _, val = (s.strip() for s in item.partition("="))
            nxt = int(val, 0) if val else nxt
            vals[name] = nxt
            nxt += 1
        out[m.group(3)] = vals
    return out


REF_SRC = "/root/reference/minivideo/src"


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="reference checkout not present")
def test_public_enums_have_the_reference_values(tmp_path):
    """every enum constant include/minivideo.h declares exists in the reference's public headers with the same value
    (minivideo.h:42-52 error codes, avcodecs.h picture / container / codec enums, avutils.h:143-149 extraction modes, ...);
    the three enums a thumbnailer passes or could receive are complete.  The values are asserted by the COMPILER on our header
    (static_assert in a translation unit), the reference's are read from its headers as text (they cannot be compiled here:
    they include a CMake-generated file)."""
    ref = {}
    for h in ("minivideo.h", "avcodecs.h", "avutils.h", "bitstream_map_struct.h", "mediafile_struct.h"):
        ref.update(_enums(os.path.join(REF_SRC, h)))
    ours = _enums(os.path.join(ROOT, "include", "minivideo.h"))
    assert {"MiniVideoErrorCodes_e", "PictureFormat_e", "PictureRepartition_e"} <= set(ours)
    lines = ['#include "minivideo.h"']
    for enum, vals in ours.items():
        assert enum in ref, enum
        for name, _ in vals.items():
            assert name in ref[enum], (enum, name)
            lines.append(f'static_assert({name} == {ref[enum][name]}, "{enum}::{name}");')
    for enum in ("MiniVideoErrorCodes_e", "PictureFormat_e", "PictureRepartition_e"):
        assert set(ours[enum]) == set(ref[enum]), (enum, set(ours[enum]) ^ set(ref[enum]))
    assert ref["MiniVideoErrorCodes_e"] == {"ERROR_UNKNOWN": 1, "ERROR_CONTAINER_UNKNOWN": 10, "ERROR_CONTAINER_FAILURE": 11,
                                            "ERROR_CODEC_UNKNOWN": 20, "ERROR_CODEC_FAILURE": 21}
    lines.append("int main() { return 0; }")
    src = tmp_path / "enums.cpp"
    src.write_text("\n".join(lines) + "\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)])
