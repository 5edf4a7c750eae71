"""minivideo_amd -- MI355X-native H.264 intra (IDR) decode path behind MiniVideo's API.

Python here is only the host-side mirror of the C interfaces (ctypes over the
C-ABI of ``libminivideo.so``).  The product path is: host entropy decode (C++)
-> packed macroblock records -> hand-written HIP kernels (gfx950).  There is no
CPU fallback for reconstruction: if the shared library or a HIP device is
missing, the calls raise.
"""
from .hotpath import (  # noqa: F401
    HotPath,
    Engine,
    PlacedBuffers,
    StreamParams,
    MiniVideoError,
    lib,
    lib_path,
    MB_BYTES,
    SUCCESS,
    FAILURE,
    UNSUPPORTED,
)

__all__ = ["HotPath", "Engine", "PlacedBuffers", "StreamParams", "MiniVideoError", "lib", "lib_path", "MB_BYTES"]
