#!/usr/bin/env python3
"""Kernel time per layout and batch size (one process, one set of buffers): where pick_layout()'s thresholds belong.
usage (GPU box, repo root): python tools/layout_crossover.py [--profile baseline|high] [--sizes 256,512,...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="baseline")
ap.add_argument("--sizes", default="256,384,512,768,1024,1536,2048")
ap.add_argument("--mbs", default="120x68")
ap.add_argument("--layouts", default="rows,quad,oct,wide,quad_wide")   # name or name:waves (waves per workgroup / rows per band)
args = ap.parse_args()
wm, hm = (int(v) for v in args.mbs.split("x"))
sizes = [int(v) for v in args.sizes.split(",")]
dev = torch.device("cuda", 0)
params, rec = synth_packed(wm, hm, 16, seed=1000, profile=args.profile, density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
Fmax = max(sizes)
d_packed = small.repeat((Fmax + 15) // 16, 1)[:Fmax].contiguous()
d_yuv = torch.empty(Fmax * params.yuv_bytes, dtype=torch.uint8, device=dev)
d_rgb = torch.empty(Fmax * params.rgb_bytes, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(dev)
hot = HotPath(0)
hot.set_fused_color(True)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream
layouts = args.layouts.split(",")
print("%-8s " % "pictures" + " ".join("%12s" % l for l in layouts) + "   (ms per launch; MB/s x 1e9 in brackets)", flush=True)
for F in sizes:
    row = []
    for layout in layouts:
        try:
            hot.set_layout(layout.split(":")[0])
            hot.set_waves_per_picture(int(layout.split(":")[1]) if ":" in layout else 0)
            for _ in range(2):
                hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            reps = 20 if F <= 256 else 5
            for _ in range(reps):
                hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3)
            e1.record(st)
            torch.cuda.synchronize(dev)
            hot.sync_check(sp)
            ms = e0.elapsed_time(e1) / reps
            row.append("%6.3f[%4.2f]" % (ms, F * params.mbs / ms / 1e6))
        except Exception as ex:   # a layout that does not fit this picture size
            row.append("     n/a    ")
    print("%-8d %s" % (F, " ".join(row)), flush=True)
hot.close()
