import sys, numpy as np
sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader
h = HotPath(0); h.set_layout("quad")
for (W,H,n) in [(1,1,3),(2,1,3),(3,2,3),(5,3,3)]:
    params, rec = synth_packed(W, H, n, seed=W*100+H, profile="baseline", density="dense")
    print("run", W, H, n, flush=True)
    g,gr = h.recon_host(params, rec, n, want_rgb=True); o,orr = loader.recon(params, rec, n, want_rgb=True)
    print(W,H,n, np.array_equal(g,o), np.array_equal(gr,orr), flush=True)
