"""CPU: the C-ABI libraries load and export every symbol declared in include/*.h (no compute calls)."""
import ctypes as C
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        names += re.findall(r"(?:MVHP_EXPORT|minivideo_EXPORT)\s+[\w\s\*]+?\b(\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_all_declared_symbols():
    from minivideo_amd import lib_path
    L = C.CDLL(lib_path())
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the reconstruction context must refuse to exist (no silent CPU path)."""
    from minivideo_amd import lib
    L = lib()
    if L.mvhp_device_count() > 0:
        return
    h = C.c_void_p()
    assert L.mvhp_create(0, C.byref(h)) == 0
    assert b"no HIP device" in L.mvhp_last_error()
