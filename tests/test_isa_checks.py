"""CPU: static checks on the compiled gfx950 ISA of the quad-layout kernel.  Its record prefetch is issued with inline
assembly and guarded by a counted s_waitcnt; the count is only right while exactly that many stores sit between a
prefetch and its use, and while nothing touches the prefetch registers in between (tools/check_prefetch_hazard.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_quad_prefetch_registers_untouched_and_store_count(tmp_path):
    _check_one(tmp_path, "recon_quad")
    _check_one(tmp_path, "recon_oct")


def _check_one(tmp_path, name):
    out = tmp_path / (name + ".s")
    src = os.path.join(ROOT, "minivideo_amd", "csrc", "hip", name + ".hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.dirname(src), src, "--cuda-device-only", "-S", "-o", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_prefetch_hazard as chk
    chk.main(str(out))
    text = out.read_text()
    assert "flat_load" not in text and "flat_store" not in text      # LDS counters must be ds_ operations
    # (register spills inside the macroblock loop would be vector-memory operations the counted waits do not know
    # about: check_prefetch_hazard rejects them; a spill of a loop-invariant outside that loop is harmless)
