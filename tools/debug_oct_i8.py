"""debug aid: where the eight-picture kernel's luma differs from the oracle on High content (macroblock, its kind and modes)."""
import sys, numpy as np
sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader
h = HotPath(0); h.set_layout("oct")
for (W, H, n) in [(6, 4, 8), (20, 12, 8)]:
    params, rec = synth_packed(W, H, n, seed=16, profile="high", density="dense")
    g, _ = h.recon_host(params, rec, n); o, _ = loader.recon(params, rec, n)
    g = g.reshape(n, -1); o = o.reshape(n, -1)
    rec = rec.reshape(n, W * H, 800)
    cnt = 0
    for f in range(n):
        Y_g = g[f, :W * H * 256].reshape(H * 16, W * 16); Y_o = o[f, :W * H * 256].reshape(H * 16, W * 16)
        for my in range(H):
            for mx in range(W):
                a = Y_g[my*16:(my+1)*16, mx*16:(mx+1)*16]; b = Y_o[my*16:(my+1)*16, mx*16:(mx+1)*16]
                if not np.array_equal(a, b):
                    r = rec[f, my * W + mx]
                    if cnt < 6:
                        bad = np.argwhere(a != b)
                        print("pic", f, "mb", (mx, my), "kind", int(r[0]), "modes", list(r[12:16]), "first bad (y,x)", bad[:6].tolist(), "got", a[tuple(bad[0])], "want", b[tuple(bad[0])])
                    cnt += 1
    print(W, H, "bad macroblocks", cnt)
