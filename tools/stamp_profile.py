#!/usr/bin/env python3
"""Where the cycles of a macroblock step go inside the eight-picture kernel: a -DMVHP_STAMPS build (tools/build_variant.sh
stamps -DMVHP_STAMPS) reads the shader clock at every section boundary of recon_oct.hip and sums the intervals per wave.
MEASUREMENT TOOL -- the stamped kernel is slower than the product (every boundary drains the wave's LDS queue).
usage (GPU box, repo root): python tools/stamp_profile.py [--profile baseline|high] [--kinds P16,P8] [--distinct N] [--no-rgb]"""
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
ap.add_argument("--lib", default="abl_tmp/stamps/libminivideo.so")
ap.add_argument("--profile", default="baseline")
ap.add_argument("--kinds", default="")
ap.add_argument("--distinct", type=int, default=16)
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--mbs", default="120x68")
ap.add_argument("--no-rgb", action="store_true")
args = ap.parse_args()
wm, hm = (int(v) for v in args.mbs.split("x"))
F = args.frames
kinds = tuple(float(v) for v in args.kinds.split(",")) if args.kinds else None
params, rec = synth_packed(wm, hm, args.distinct, seed=1000, profile=args.profile, density="dense", kinds=kinds, qp_range=(24, 32))
dev = torch.device("cuda", 0)
small = torch.from_numpy(rec.reshape(args.distinct, -1)).to(dev)
d_packed = small.repeat((F + args.distinct - 1) // args.distinct, 1)[:F].contiguous()
d_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
d_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
L = C.CDLL(os.path.join(R, args.lib))
vp, i32 = C.c_void_p, C.c_int
L.mvhp_create.argtypes = [i32, C.POINTER(vp)]
L.mvhp_recon_stages_dev.argtypes = [vp, C.POINTER(StreamParams), vp, i32, vp, vp, vp, i32]
L.mvhp_set_layout.argtypes = [vp, i32]
L.mvhp_sync_check.argtypes = [vp, vp]
L.mvhp_debug_read_stamps.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(i32)]
h = vp()
assert L.mvhp_create(0, C.byref(h)) == 1
L.mvhp_set_layout(h, 3)
st = torch.cuda.Stream(device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(2):
    e0.record(st)
    assert L.mvhp_recon_stages_dev(h, C.byref(params), d_packed.data_ptr(), F, d_yuv.data_ptr(),
                                   None if args.no_rgb else d_rgb.data_ptr(), st.cuda_stream, 1 if args.no_rgb else 3) == 1
    e1.record(st)
    torch.cuda.synchronize()
assert L.mvhp_sync_check(h, st.cuda_stream) == 1
ms = e0.elapsed_time(e1)
names = (C.c_char_p * 64)()
cnt = i32()
buf = (C.c_uint32 * (256 * 8 * 32))()
assert L.mvhp_debug_read_stamps(buf, names, C.byref(cnt)) == 1
a = np.frombuffer(buf, np.uint32).reshape(256, 8, 32).astype(np.float64)
n = cnt.value
wg = min(256, (F + 7) // 8)
steps_per_wg = wm * hm
tot = a[:wg, :, :n].sum(axis=(0, 1))                 # cycles summed over every wave of every workgroup
per_step = tot / (wg * steps_per_wg)                  # average cycles a wave spends per macroblock step
# the clock of s_memtime: one wave's total against the launch time
wave_total = a[:wg, :, :n].sum(axis=2)
print("stamped launch %.3f ms; a wave's stamped intervals sum to %.3g ticks (max over waves) -> %.1f MHz tick" %
      (ms, wave_total.max(), wave_total.max() / (ms * 1e-3) / 1e6))
print("%-16s %10s %7s" % ("interval ends at", "ticks/step", "share"))
for i in range(n):
    print("%-16s %10.1f %6.1f%%" % (names[i].decode(), per_step[i], 100 * per_step[i] / per_step.sum()))
print("%-16s %10.1f" % ("sum", per_step.sum()))
