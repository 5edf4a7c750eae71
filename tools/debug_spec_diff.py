#!/usr/bin/env python3
"""Developer aid: first macroblocks where a layout's pictures differ from the oracle's on a generated MVHP_STREAM_SPEC stream
(several slices / I_PCM / scaling lists).  usage (GPU box): python tools/debug_spec_diff.py profile slices pcm scaling [layout]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from minivideo_amd import HotPath, gen
from oracle import loader
from tests.util import Stream
prof, slices, pcm, sc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
layout = sys.argv[5] if len(sys.argv) > 5 else "rows"
W, H, F = 11, 7, 5
stream, packed, _ = gen.make_stream_ex(W, H, F, seed=77 + slices + pcm, profile=prof, slices=slices, pcm_permille=pcm, scaling=sc, qp_range=(10, 45))
with Stream(stream, spec=True) as s:
    p = s.params(0)
hot = HotPath(0); hot.set_layout(layout)
yuv, _ = hot.recon_host(p, packed, F)
ref, _ = loader.recon(p, packed, F)
fb = W * H * 384
n = 0
for f in range(F):
    a, b = yuv[f * fb:(f + 1) * fb], ref[f * fb:(f + 1) * fb]
    Y, Yr = a[:W * H * 256].reshape(H * 16, W * 16), b[:W * H * 256].reshape(H * 16, W * 16)
    Cb, Cbr = a[W * H * 256:W * H * 320].reshape(H * 8, W * 8), b[W * H * 256:W * H * 320].reshape(H * 8, W * 8)
    for mb in range(W * H):
        x, y = mb % W, mb // W
        dy = (Y[y * 16:y * 16 + 16, x * 16:x * 16 + 16] != Yr[y * 16:y * 16 + 16, x * 16:x * 16 + 16])
        dc = (Cb[y * 8:y * 8 + 8, x * 8:x * 8 + 8] != Cbr[y * 8:y * 8 + 8, x * 8:x * 8 + 8])
        if dy.any() or dc.any():
            h = packed[f, mb, :32]
            print("pic %d mb %d (x %d y %d): kind %d qp %d cmode %d i16mode %d unavail %x modes %s | luma diff rows %s cols %s chroma diff %s" % (
                f, mb, x, y, h[0], h[1], h[3], h[4], h[6], list(h[12:28]), sorted(set(np.nonzero(dy)[0])), sorted(set(np.nonzero(dy)[1])), bool(dc.any())))
            n += 1
            if n >= 6: sys.exit(0)
print("differences:", n)
