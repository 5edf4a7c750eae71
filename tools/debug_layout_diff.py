import sys, numpy as np
sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader
h = HotPath(0); h.set_layout("quad")
for (W,H,prof) in [(11,9,"baseline"),(20,17,"baseline"),(11,9,"high")]:
    params, rec = synth_packed(W, H, 3, seed=W*100+H, profile=prof, density="dense")
    g,_ = h.recon_host(params, rec, 3); o,_ = loader.recon(params, rec, 3)
    fb = params.yuv_bytes
    bad = np.nonzero(g != o)[0]
    print(W,H,prof,"bad",bad.size)
    seen = {}
    for b in bad:
        f, off = divmod(int(b), fb)
        if off < W*H*256:
            y, x = divmod(off, W*16); key=(f,'Y',x//16,y//16)
        else:
            off2 = off - W*H*256; pl = off2 // (W*H*64); off2 %= W*H*64
            y, x = divmod(off2, W*8); key=(f,'C%d'%pl,x//8,y//8)
        seen.setdefault(key, []).append((x%16 if key[1]=='Y' else x%8, y%16 if key[1]=='Y' else y%8, int(g[b]), int(o[b])))
    for k in sorted(seen)[:12]:
        f,pl,mx,my = k
        r = rec[f, my*W+mx]
        print(k, "kind",r[0],"qp",r[1],"cbp",r[2],"cmode",r[3],"i16",r[4],"nz",hex(int(r[8:12].view(np.uint32)[0])), "n",len(seen[k]), seen[k][:6])
