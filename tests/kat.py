"""Known-answer vectors whose expected outputs are REFERENCE outputs recorded in
SURVEY.md Appendix A (the survey session ran the real MiniVideo decoder on
these two-macroblock streams).  Here they are restated at packed-record level:
MB0 = Intra16x16 DC prediction with one luma DC level +3, MB1 = Intra16x16
Horizontal / chroma Horizontal, nothing else coded, picture 32x16."""
import numpy as np

from minivideo_amd.hotpath import StreamParams

# slice QP -> luma value the reference produced (36 is the
# h264_transform.c:797-808 `qP > 36` defect)
EXPECTED_Y = {28: 131, 35: 135, 36: 0, 37: 136}
EXPECTED_YUV_MD5_QP28 = "2d87b01fcabfeea0afc04f99c5b30083"
EXPECTED_RGB_QP28 = (134, 133, 134)


def kat_packed(qp):
    rec = np.zeros((1, 2, 800), np.uint8)
    for mb, (i16, cm) in enumerate([(2, 0), (1, 1)]):
        rec[0, mb, 0] = 2        # Intra16x16
        rec[0, mb, 1] = qp
        rec[0, mb, 3] = cm       # chroma: DC / Horizontal
        rec[0, mb, 4] = i16      # luma: DC / Horizontal
    rec[0, 0, 32:].view(np.int16)[0] = 3
    rec[0, 0, 8:12] = np.array([1], np.uint32).view(np.uint8)
    return StreamParams(2, 1, 0, 0, 0), rec
