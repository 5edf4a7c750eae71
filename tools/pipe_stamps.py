#!/usr/bin/env python3
"""Where a step of the prediction wave (K) of recon_pipe_kernel goes: a -DMVHP_PIPE_STAMPS build reads the shader clock at the
section boundaries of K's macroblock loop and sums the intervals over the launch.  MEASUREMENT TOOL (every boundary drains the
wave's LDS queue: the stamped kernel is slower than the product).
  bash tools/build_variant.sh pstamps -DMVHP_PIPE_STAMPS && python tools/pipe_stamps.py [--profile baseline|high] [--frames 16,64]"""
import argparse
import ctypes as C
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MINIVIDEO_LIB"] = os.path.join(R, "abl_tmp", "pstamps", "libminivideo.so")
from minivideo_amd import HotPath, lib
from minivideo_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="baseline")
ap.add_argument("--frames", default="1,16,64")
ap.add_argument("--rows", default="4")
args = ap.parse_args()
NAMES = ["seam", "wait O / F", "header + wait row above + top fetch", "chroma", "luma", "neighbours + publish"]
L = lib()
L.mvhp_debug_pipe_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
params, rec = synth_packed(120, 68, 16, seed=1000, profile=args.profile, density="dense")
dev = torch.device("cuda", 0)
small = torch.from_numpy(rec.reshape(16, -1)).to(dev)
hot = HotPath(0)
hot.set_layout("pipe")
st = torch.cuda.Stream(device=dev)
for rows in [int(v) for v in args.rows.split(",")]:
    hot.set_waves_per_picture(rows)
    for F in [int(v) for v in args.frames.split(",")]:
        d_packed = small.repeat((F + 15) // 16, 1)[:F].contiguous()
        d_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
        d_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hot.recon_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), st.cuda_stream)
        hot.sync_check(st.cuda_stream)
        L.mvhp_debug_pipe_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        hot.recon_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), d_rgb.data_ptr(), st.cuda_stream)
        e1.record(st)
        hot.sync_check(st.cuda_stream)
        buf = (C.c_ulonglong * 32)()
        assert L.mvhp_debug_pipe_stamps(buf, 0) == 1
        steps = max(1, buf[2 * 5 + 1])
        tot = sum(buf[2 * i] for i in range(6))
        print("%s, %d pictures, %d rows per band: %.3f ms (stamped build); %d K steps, %.0f clock ticks per step" % (
            args.profile, F, rows, e0.elapsed_time(e1), steps, tot / steps), flush=True)
        for i, nm in enumerate(NAMES):
            print("    %-40s %8.0f ticks per step  %5.1f %%" % (nm, buf[2 * i] / steps, 100.0 * buf[2 * i] / max(1, tot)))
        del d_packed, d_yuv, d_rgb
hot.close()
