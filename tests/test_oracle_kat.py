"""CPU: the oracle restatement against the reference outputs recorded in SURVEY.md Appendix A."""
import hashlib

import numpy as np
import pytest

from oracle import loader
from tests.kat import EXPECTED_RGB_QP28, EXPECTED_Y, EXPECTED_YUV_MD5_QP28, kat_packed


@pytest.mark.parametrize("qp", sorted(EXPECTED_Y))
def test_oracle_matches_reference_kat(qp):
    params, rec = kat_packed(qp)
    yuv, rgb = loader.recon(params, rec, 1, want_rgb=True)
    assert np.all(yuv[:512] == EXPECTED_Y[qp])
    assert np.all(yuv[512:] == 128)
    if qp == 28:
        assert hashlib.md5(yuv.tobytes()).hexdigest() == EXPECTED_YUV_MD5_QP28
        assert np.all(rgb.reshape(-1, 3) == np.array(EXPECTED_RGB_QP28, np.uint8))


def test_colour_formula_hand_values():
    # export_utils.c:300-302 on a few hand-computed triples
    from minivideo_amd.hotpath import StreamParams
    import ctypes as C
    L = loader.lib()
    p = StreamParams(1, 1, 0, 0, 0)
    yuv = np.zeros(384, np.uint8)
    yuv[:256] = 255; yuv[256:320] = 0; yuv[320:] = 255
    rgb = np.zeros(768, np.uint8)
    L.orc_yuv_to_rgb(C.byref(p), yuv.ctypes.data, rgb.ctypes.data)
    r = ((298 * 255) >> 8) + ((408 * 255) >> 8) - 222
    g = ((298 * 255) >> 8) - 0 - ((208 * 255) >> 8) + 135
    b = ((298 * 255) >> 8) + 0 - 276
    exp = [min(max(v, 0), 255) for v in (r, g, b)]
    assert rgb.reshape(-1, 3)[0].tolist() == exp
