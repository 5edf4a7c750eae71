import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=["rows", "quad", "oct", "wide", "quad_wide", "pipe"])
def hot(request):
    """GPU reconstruction context through the C-ABI (fails loudly without libminivideo.so / a GPU), once per
    kernel layout: one ("rows"), four ("quad") and eight ("oct") pictures per workgroup, and one picture / four pictures over several
    workgroups ("wide", "quad_wide")."""
    from minivideo_amd import HotPath
    h = HotPath(0)
    h.set_layout(request.param)
    yield h
    h.close()


def pytest_collection_modifyitems(config, items):
    """The placement tests go FIRST.  They hold and free arenas of up to 200 GB; the driver wipes released device memory in the
    background (about 3 s for 200 GB), and a process that exits with that still pending makes the NEXT process on the device --
    the driver's smoke() and bench.py -- wait for it in its first allocations (a 3.8-s "cold call", profiles/r03m_base1080_bench.json).
    At the front of the suite the wipe is long over when pytest exits."""
    front = [it for it in items if "test_gpu_placement" in it.nodeid]
    if front:
        rest = [it for it in items if "test_gpu_placement" not in it.nodeid]
        items[:] = front + rest
