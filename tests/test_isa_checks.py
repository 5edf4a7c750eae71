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


def test_checker_catches_violations(tmp_path):
    """The safety net itself: inject (a) a compiler-style instruction that touches a prefetch register while the loads
    are in flight, (b) an extra asm store into the loop, (c) a scratch access into the loop -- each must be rejected."""
    import re
    out = tmp_path / "recon_quad.s"
    src = os.path.join(ROOT, "minivideo_amd", "csrc", "hip", "recon_quad.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.dirname(src), src, "--cuda-device-only", "-S", "-o", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_prefetch_hazard as chk
    lines = out.read_text().split("\n")
    # the last asm load block of the first kernel = the prefetch inside the macroblock loop
    k0 = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4mvhp17recon_quad_kernel\S+:", l))
    k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
    loads = [i for i in range(k0, k1) if "global_load_dwordx4 v[120:123]" in lines[i]]
    at = loads[-1] + 2          # behind the #ASMEND of the block
    for inject in ("\tv_mov_b32_e32 v101, v1",
                   "\t;;#ASMSTART\n\ts_nop 4\n\tglobal_store_dwordx4 v1, v[2:5], s[0:1]\n\ts_nop 1\n\t;;#ASMEND",
                   "\tscratch_load_dword v1, off, off"):
        bad = lines[:at] + inject.split("\n") + lines[at:]
        p = tmp_path / "bad.s"
        p.write_text("\n".join(bad))
        with pytest.raises(chk.HazardError):
            chk.main(str(p))
    # (d) the hand-padded hazards: the round-1 GPU memory fault was a vector-memory instruction reading
    # its scalar base right behind the v_readlane that wrote it.  Drop the wait states in front of the loop's prefetch
    # block, or behind a 128-bit store, and the checker must refuse the ISA.
    blk0 = max(i for i in range(k0, loads[-1]) if "#ASMSTART" in lines[i])
    nop = next(i for i in range(blk0, loads[-1]) if re.match(r"\s*s_nop 4", lines[i]))
    p = tmp_path / "bad.s"
    p.write_text("\n".join(lines[:nop] + lines[nop + 1:]))
    with pytest.raises(chk.HazardError, match="wait states in front"):
        chk.main(str(p))
    st = next(i for i in range(k0, k1) if re.match(r"\s*global_store_dwordx4", lines[i]) and re.match(r"\s*s_nop 4", lines[i - 1]))
    assert re.match(r"\s*s_nop 1", lines[st + 1])   # (an inline-assembly store of the strip flush)
    p.write_text("\n".join(lines[:st + 1] + lines[st + 2:]))
    with pytest.raises(chk.HazardError, match="wait states behind"):
        chk.main(str(p))
