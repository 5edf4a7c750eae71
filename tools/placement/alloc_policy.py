"""Does it matter HOW a batch's three buffers are allocated (ordinary hipMalloc through torch)?  Config 2's launch on
  consecutive: records, planes, RGB one after the other (what a program does without thinking)
  spaced:      a spacer of S GB between records / planes / RGB while they are allocated (freed afterwards)
each from an emptied allocator cache, several times in one process.  usage: python tools/placement/alloc_policy.py [S GB] [trials]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed

S = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
F = 2048
params, rec = synth_packed(120, 68, 16, seed=1000, profile="baseline", density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
hot = HotPath(0); hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev); sp = st.cuda_stream

def alloc(n): return torch.empty(int(n), dtype=torch.uint8, device=dev)

def run(policy, junk_gb):
    torch.cuda.empty_cache()
    junk = alloc(junk_gb * 1e9) if junk_gb else None     # whatever the process allocated before (moves the start)
    sa = sb = None
    d_packed = small.repeat(F // 16, 1).contiguous()
    if policy == "spaced": sa = alloc(S * 1e9)
    d_yuv = alloc(F * params.yuv_bytes)
    if policy == "spaced": sb = alloc(S * 1e9)
    d_rgb = alloc(F * params.rgb_bytes)
    del sa, sb
    torch.cuda.synchronize()
    def go(): hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): go()
    e1.record(st); torch.cuda.synchronize(); hot.sync_check(sp)
    ms = e0.elapsed_time(e1) / 5
    del d_packed, d_yuv, d_rgb, junk
    return ms

print("spacer %.0f GB; ms per launch (2048 x 1080p Baseline)" % S)
print("%-8s %12s %12s" % ("junk GB", "consecutive", "spaced"))
for t in range(trials):
    junk = [0, 7, 19, 33, 50, 71, 90][t % 7]
    a = run("consecutive", junk); b = run("spaced", junk)
    print("%-8d %12.3f %12.3f" % (junk, a, b), flush=True)
