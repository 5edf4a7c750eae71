import sys, numpy as np
sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader
h = HotPath(0); h.set_layout("quad")
for waves in (8, 12):
    h.set_waves_per_picture(waves)
    for (W,H,prof,n) in [(23,37,"high",5),(24,37,"high",5),(23,37,"baseline",5),(8,30,"baseline",4)]:
        params, rec = synth_packed(W, H, n, seed=16, profile=prof, density="dense")
        g,gr = h.recon_host(params, rec, n, want_rgb=True); o,orr = loader.recon(params, rec, n, want_rgb=True)
        bad = np.nonzero(g != o)[0]; badr = np.nonzero(gr != orr)[0]
        print("waves",waves,W,H,prof,"yuv bad",bad.size,"rgb bad",badr.size)
        fb = params.rgb_bytes
        seen = {}
        for b in badr[:4000]:
            f, off = divmod(int(b), fb)
            y, x = divmod(off//3, W*16)
            seen.setdefault((f, x//16, y//16), []).append((x%16,y%16,off%3,int(gr[b]),int(orr[b])))
        for k in sorted(seen)[:8]:
            print("  ", k, len(seen[k]), seen[k][:5])
