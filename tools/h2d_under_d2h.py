#!/usr/bin/env python3
"""Why the engine's H2D stage reports 8 GB/s (BENCH_r03 stages_rank0.h2d) when a compact picture crosses the link at 25+ GB/s on
an idle link: the same ten 0.88-MB page-locked pieces per chunk (decode_engine.cpp uploader -> eng_h2d), timed with HIP events
on their stream, (a) alone, (b) while another stream downloads 9.4-MB pictures back to back (the engine's D2H stage, which is
busy 2/3 of a Baseline job), (c) as ONE contiguous copy of the same bytes, alone and under the same download load.
usage (GPU box): python tools/h2d_under_d2h.py"""
import torch

dev = torch.device("cuda", 0)
PIECE, N = 880_000, 10
PIC_OUT = 9_400_320
h_in = torch.empty(N * PIECE, dtype=torch.uint8).pin_memory()
d_in = torch.empty(N * PIECE, dtype=torch.uint8, device=dev)
h_out = torch.empty(8 * PIC_OUT, dtype=torch.uint8).pin_memory()
d_out = torch.empty(8 * PIC_OUT, dtype=torch.uint8, device=dev)
up, down = torch.cuda.Stream(), torch.cuda.Stream()


def upload(pieces):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(up):
        e0.record()
        if pieces:
            for i in range(N):
                d_in[i * PIECE:(i + 1) * PIECE].copy_(h_in[i * PIECE:(i + 1) * PIECE], non_blocking=True)
        else:
            d_in.copy_(h_in, non_blocking=True)
        e1.record()
    return e0, e1


def run(pieces, loaded, reps=40):
    times = []
    for _ in range(reps):
        if loaded:
            with torch.cuda.stream(down):
                for _ in range(3):
                    h_out.copy_(d_out, non_blocking=True)   # 75 MB per copy: ~4 ms of download around the upload
        e0, e1 = upload(pieces)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    ms = times[len(times) // 2]
    return ms, N * PIECE / ms / 1e6


for pieces in (True, False):
    for loaded in (False, True):
        ms, gbs = run(pieces, loaded)
        print("%-28s %-26s %.3f ms per chunk of %d x %.2f MB = %.1f GB/s" % (
            "ten copies (as the engine)" if pieces else "one contiguous copy", "while D2H runs" if loaded else "link otherwise idle",
            ms, N, PIECE / 1e6, gbs), flush=True)

# (d) the same ten pieces as ONE hipMemcpy2DAsync: rows = pictures, pitch = the slot size of a picture in the page-locked chunk and
#     in the device staging buffer (6.5 MB), width = the largest compact picture of the chunk
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
PITCH = 6_560_000
h2 = torch.empty(N * PITCH, dtype=torch.uint8).pin_memory()
d2 = torch.empty(N * PITCH, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for loaded in (False, True):
    times = []
    for _ in range(40):
        if loaded:
            with torch.cuda.stream(down):
                for _ in range(3):
                    h_out.copy_(d_out, non_blocking=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(up):
            e0.record()
            rc = hip.hipMemcpy2DAsync(d2.data_ptr(), PITCH, h2.data_ptr(), PITCH, PIECE, N, 1, C.c_void_p(up.cuda_stream))
            assert rc == 0, rc
            e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    ms = times[len(times) // 2]
    print("%-28s %-26s %.3f ms per chunk of %d x %.2f MB = %.1f GB/s" % ("one hipMemcpy2DAsync", "while D2H runs" if loaded else "link otherwise idle",
                                                                       ms, N, PIECE / 1e6, N * PIECE / ms / 1e6), flush=True)
