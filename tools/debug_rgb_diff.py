"""Where does a build's RGB differ from the oracle?  usage: MINIVIDEO_LIB=... python tools/debug_rgb_diff.py [layout]"""
import sys, numpy as np
sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader
h = HotPath(0); h.set_layout(sys.argv[1] if len(sys.argv) > 1 else "oct")
for (W, H, prof, n) in [(8, 2, "baseline", 1), (12, 3, "baseline", 2)]:
    params, rec = synth_packed(W, H, n, seed=16, profile=prof, density="dense")
    g, gr = h.recon_host(params, rec, n, want_rgb=True); o, orr = loader.recon(params, rec, n, want_rgb=True)
    bad = np.nonzero(g != o)[0]; badr = np.nonzero(gr != orr)[0]
    print(W, H, prof, "yuv bad", bad.size, "rgb bad", badr.size, "of", gr.size)
    fb = params.rgb_bytes
    rows = {}
    for b in badr:
        f, off = divmod(int(b), fb)
        y, xb = divmod(off, W * 48)
        rows.setdefault((f, y), []).append(xb)
    for k in sorted(rows)[:6]:
        xs = rows[k]
        print("  picture %d row %d: %d bad bytes, 16-byte pieces %s" % (k[0], k[1], len(xs), sorted({x // 16 for x in xs})))
    # is a bad piece some other piece of the oracle row?
    if badr.size:
        f, y = sorted(rows)[0]
        grow = gr[f * fb + y * W * 48: f * fb + (y + 1) * W * 48].reshape(-1, 16)
        orow = orr[f * fb + y * W * 48: f * fb + (y + 1) * W * 48].reshape(-1, 16)
        for q in range(grow.shape[0]):
            src = [k for k in range(orow.shape[0]) if np.array_equal(grow[q], orow[k])]
            print("   piece", q, "holds oracle piece", src)
