#!/usr/bin/env python3
"""A/B timing of builds of libminivideo.so (tools/build_variant.sh) inside ONE process on the SAME device buffers.

Why: the time of the 1080p Baseline launch depends on where its buffers were placed (tools/placement/alloc_variance.py: 9.1 ... 12.4 ms
for one binary in one process), so timings from different processes, let alone boxes, do not compare builds.  Here every
build is loaded side by side (RTLD_LOCAL), the buffers are re-allocated `--trials` times, and every build runs on each set.
usage (GPU box, repo root): python tools/ab_same_buffers.py product=minivideo_amd/libminivideo.so x=abl_tmp/x/libminivideo.so
       [--profile baseline|high] [--frames 2048] [--trials 8] [--launches 6] [--mbs 120x68]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from minivideo_amd.hotpath import StreamParams
from minivideo_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+", help="name=path")
ap.add_argument("--profile", default="baseline")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--trials", type=int, default=8)
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--mbs", default="120x68")
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--shuffle", type=int, default=0, help="1: picture f = a random one of the 16 (seeded) instead of f % 16")
args = ap.parse_args()
wm, hm = (int(v) for v in args.mbs.split("x"))
dev = torch.device("cuda", 0)
F = args.frames
params, rec = synth_packed(wm, hm, 16, seed=1000, profile=args.profile, density="dense")
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
st = torch.cuda.Stream(device=dev)
sp = st.cuda_stream

builds = []
for spec in args.libs:
    name, path = spec.split("=", 1)
    L = C.CDLL(os.path.join(R, path) if not os.path.isabs(path) else path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    vp, i32 = C.c_void_p, C.c_int
    L.mvhp_create.argtypes = [i32, C.POINTER(vp)]
    L.mvhp_recon_stages_dev.argtypes = [vp, C.POINTER(StreamParams), vp, i32, vp, vp, vp, i32]
    L.mvhp_set_fused_color.argtypes = [vp, i32]
    L.mvhp_set_layout.argtypes = [vp, i32]
    L.mvhp_sync_check.argtypes = [vp, vp]
    h = vp()
    assert L.mvhp_create(0, C.byref(h)) == 1, name
    L.mvhp_set_fused_color(h, 1)
    L.mvhp_set_layout(h, args.layout)
    builds.append((name, L, h))


def run(L, h, d_packed, d_yuv, d_rgb):
    def go():
        assert L.mvhp_recon_stages_dev(h, C.byref(params), d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), sp, 3) == 1
    go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.launches):
        go()
    e1.record(st)
    torch.cuda.synchronize(dev)
    assert L.mvhp_sync_check(h, sp) == 1
    return e0.elapsed_time(e1) / args.launches


hold = []
ref = None
table = {name: [] for name, _, _ in builds}
print("%-6s " % "trial" + " ".join("%10s" % n for n, _, _ in builds), flush=True)
for t in range(args.trials):
    if args.shuffle:
        g = torch.Generator(device="cpu").manual_seed(1)
        d_packed = small[torch.randint(0, 16, (F,), generator=g).to(dev)].contiguous()
    else:
        d_packed = small.repeat((F + 15) // 16, 1)[:F].contiguous()
    d_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
    d_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)   # the records are copied on torch's stream, the kernels run on `st`
    hold += [d_packed, d_yuv, d_rgb]
    if len(hold) > 15:   # keep five sets alive so that new sets land elsewhere, free the oldest
        del hold[:3]
    row = []
    for name, L, h in builds + builds[::-1]:   # A B ... B A: order effects cancel in the mean of the two
        ms = run(L, h, d_packed, d_yuv, d_rgb)
        row.append(ms)
        # every build must produce the same bytes
        sig = (int(d_yuv[::4099].to(torch.int64).sum()), int(d_rgb[::4099].to(torch.int64).sum()))
        if ref is None:
            ref = sig
        if sig != ref:
            print("OUTPUT DIFFERS for build", name, flush=True)
    n = len(builds)
    means = [(row[k] + row[2 * n - 1 - k]) / 2 for k in range(n)]
    for (name, _, _), m in zip(builds, means):
        table[name].append(m)
    print("%-6d " % t + " ".join("%10.3f" % m for m in means), flush=True)
print("%-6s " % "mean" + " ".join("%10.3f" % np.mean(table[n]) for n, _, _ in builds))
print("%-6s " % "median" + " ".join("%10.3f" % np.median(table[n]) for n, _, _ in builds))
print("%-6s " % "min" + " ".join("%10.3f" % np.min(table[n]) for n, _, _ in builds))
print("%-6s " % "max" + " ".join("%10.3f" % np.max(table[n]) for n, _, _ in builds))
