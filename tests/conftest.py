import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=["rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"])
def hot(request):
    """GPU reconstruction context through the C-ABI (fails loudly without libminivideo.so / a GPU), once per
    kernel layout: one ("rows"), four ("quad") and eight ("oct") pictures per workgroup, and one picture / four pictures over several
    workgroups ("wide", "quad_wide")."""
    from minivideo_amd import HotPath
    h = HotPath(0)
    h.set_layout(request.param)
    yield h
    h.close()
