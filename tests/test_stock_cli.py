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
