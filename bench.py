#!/usr/bin/env python3
"""bench.py -- macroblocks/s of the MI355X-native H.264 intra reconstruction hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under
torch.distributed.run, one rank per GPU).  One "step" = one pass of the hot path
(reconstruction kernel + colour kernel) over one batch of synthetic 1080p
Baseline pictures whose packed macroblock records are already resident in HBM.
Frames are independent, so ranks share nothing on the data path (weak scaling:
every rank processes its own batch); torch.distributed is used only for the
barrier and the max-over-ranks clock.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP events on the
launch stream) and, at N=1, `cpu_baseline` (the CPU restatement in oracle/,
one thread, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_MB_RECON = 800 + 384  # SURVEY.md 8(d) B_yuv: packed record in + planar YCbCr out
BYTES_PER_MB_FUSED = 800 + 384 + 768  # SURVEY.md 8(d) B_rgb: + RGB out written by the fused colour epilogue
BYTES_PER_MB_COLOR = 384 + 768  # colour kernel: YCbCr in + RGB out (B_rgb - B_yuv = 768 written)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=2048, help="pictures per GPU per step (512 four-picture workgroups = "
                    "two 8-wave workgroups on each of the 256 CUs)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pictures tiled to --frames")
    ap.add_argument("--width-mbs", type=int, default=120)
    ap.add_argument("--height-mbs", type=int, default=68)
    ap.add_argument("--profile", default="baseline", choices=["baseline", "high"])
    ap.add_argument("--density", default="dense", choices=["dense", "light"])
    ap.add_argument("--source", default="stream", choices=["stream", "records"],
                    help="stream: synthetic Annex-B stream -> host front end -> packed records (default); "
                         "records: random packed records drawn directly (minivideo_amd.synth)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--layout", default="auto", choices=["auto", "rows", "quad", "oct"],
                    help="pictures per workgroup: rows = 1 (one wavefront per macroblock row), quad = 4 (16 lanes per picture)")
    ap.add_argument("--no-rgb", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="run the colour conversion as its own kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


PMC_SUMMARY = "r01_v6_pmc_summary.json"


def measured_traffic(args, fused, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command
    (tools/pmc_profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950
    correction in MI355X_MICROARCH.md).  Only reported when the run uses the profiled configuration."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    default = (args.frames == 2048 and args.width_mbs == 120 and args.height_mbs == 68 and args.profile == "baseline"
               and args.density == "dense" and args.source == "stream" and fused and not args.waves and args.layout == "auto")
    if not (default and os.path.exists(path)):
        return None, None
    d = json.load(open(path)).get(kernel.split(" ")[0], {})
    if "hbm_read_bytes_corrected" not in d or "hbm_write_bytes" not in d:
        return None, None
    return d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"], "profiles/" + PMC_SUMMARY


def cpu_baseline(params, rec, want_rgb, budget_s):
    """Time the CPU restatement (oracle/, kind "port") on a bounded sample of the same workload."""
    from oracle import loader
    n_have = rec.shape[0]
    loader.recon(params, rec[:1], 1, want_rgb=want_rgb)  # warm
    n = 0
    t0 = time.perf_counter()
    while True:
        loader.recon(params, rec, n_have, want_rgb=want_rgb)
        n += n_have
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {
        "value": n * params.mbs / dt,
        "unit": "macroblocks/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} of the benchmark's 1080p pictures ({n * params.mbs} macroblocks, {dt:.1f} s), "
                  f"oracle/recon_ref.c single thread, same stages as the GPU step",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the reconstruction hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from minivideo_amd import HotPath

    want_rgb = not args.no_rgb
    F = args.frames
    n_distinct = min(args.distinct, F)
    host_rate = None
    if args.source == "stream":
        # the real input path: Annex-B bytes -> host entropy decode (CAVLC / CABAC) -> packed records
        import ctypes as C
        from minivideo_amd import gen, lib
        from minivideo_amd.hotpath import StreamParams
        stream, _ = gen.make_stream(args.width_mbs, args.height_mbs, n_distinct, seed=1000 + rank,
                                    profile=args.profile, dense=(args.density == "dense"), want_packed=False)
        L = lib()
        h = C.c_void_p()
        if L.mvhp_stream_open(stream.ctypes.data, stream.size, C.byref(h)) != 1:
            raise SystemExit("bench: the synthetic stream failed to parse")
        params = StreamParams()
        L.mvhp_stream_params(h, 0, C.byref(params))
        rec = np.zeros((n_distinct, params.mbs, 800), np.uint8)
        t0 = time.perf_counter()
        for k in range(n_distinct):
            if L.mvhp_stream_decode_packed(h, k, rec[k].ctypes.data, rec[k].nbytes) != 1:
                raise SystemExit("bench: host front end failed on picture %d" % k)
        host_rate = n_distinct * params.mbs / (time.perf_counter() - t0)
        L.mvhp_stream_close(h)
        stream_bytes = int(stream.size)
    else:
        from minivideo_amd.synth import synth_packed
        params, rec = synth_packed(args.width_mbs, args.height_mbs, n_distinct, seed=1000 + rank,
                                   profile=args.profile, density=args.density)
        stream_bytes = None
    dev = torch.device("cuda", local_rank)
    d_small = torch.from_numpy(rec.reshape(rec.shape[0], -1)).to(dev)
    reps = (F + d_small.shape[0] - 1) // d_small.shape[0]
    d_packed = d_small.repeat(reps, 1)[:F].contiguous()
    del d_small
    d_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
    d_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev) if want_rgb else None
    torch.cuda.synchronize(dev)   # inputs are resident before anything is launched on the bench stream
    hot = HotPath(local_rank)
    if args.waves:
        hot.set_waves_per_picture(args.waves)
    hot.set_layout(args.layout)
    # which reconstruction kernel the library picks (mirrors pick_quad() in hotpath_abi.hip: speed only)
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    octl = args.layout == "oct" or (args.layout == "auto" and F >= 8 * n_cus and not (params.flags & 1))
    quad = not octl and (args.layout == "quad" or (args.layout == "auto" and F >= 3 * n_cus))
    recon_name = "recon_oct_kernel" if octl else ("recon_quad_kernel" if quad else "recon_rows_kernel")
    fused = want_rgb and not args.no_fused
    hot.set_fused_color(fused)
    # a dedicated (non-null) stream: the C-ABI treats a NULL stream as "the context's own stream",
    # and the HIP events below must sit on the stream the kernels are launched on.
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    assert sp != 0
    rgb_ptr = d_rgb.data_ptr() if want_rgb else None

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), rgb_ptr, sp, 3 if fused else 1)
        if ev is not None:
            ev[1].record(stream)
        if want_rgb and not fused:
            hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), rgb_ptr, sp, 2)
        if ev is not None:
            ev[2].record(stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    hot.sync_check(sp)

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    barrier()
    elapsed = time.perf_counter() - t0
    hot.sync_check(sp)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_recon = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    ms_color = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if (want_rgb and not fused) else 0.0

    mbs_per_step = F * params.mbs
    value = world * mbs_per_step * args.steps / elapsed

    # spot-check one picture of the last step against the oracle (bit-exact) -- the checker, not the product
    ok = None
    if rank == 0:
        from oracle import loader
        ok = True
        for f in sorted({0, 1, 2, 3, F // 2, F - 1} & set(range(F))):
            src = f % rec.shape[0]
            ref, ref_rgb = loader.recon(params, rec[src:src + 1], 1, want_rgb=want_rgb)
            ok = ok and bool(np.array_equal(d_yuv[f * params.yuv_bytes:(f + 1) * params.yuv_bytes].cpu().numpy(), ref))
            if want_rgb:
                ok = ok and bool(np.array_equal(d_rgb[f * params.rgb_bytes:(f + 1) * params.rgb_bytes].cpu().numpy(), ref_rgb))

    if rank == 0:
        dom_recon = ms_recon >= ms_color
        bpm = BYTES_PER_MB_FUSED if fused else BYTES_PER_MB_RECON
        if dom_recon:
            achieved = mbs_per_step * bpm / (ms_recon * 1e-3) / 1e9
            kname = recon_name + (" (fused RGB epilogue)" if fused else "")
        else:
            achieved = mbs_per_step * BYTES_PER_MB_COLOR / (ms_color * 1e-3) / 1e9
            kname = "ycbcr_to_rgb_kernel"
        traffic, traffic_src = measured_traffic(args, fused, kname)
        out = {
            "metric": "macroblocks/s on 1080p H.264 IDR frames; 1/2/4/8-GPU scaling",
            "value": value,
            "unit": "macroblocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": (f"synthetic ({n_distinct} distinct {args.density} pictures of a generated "
                     f"{'CAVLC' if args.profile == 'baseline' else 'CABAC'} Annex-B stream, entropy-decoded by the host front end, "
                     if args.source == "stream" else f"synthetic ({n_distinct} distinct random {args.density} pictures, ")
                    + "tiled to the batch; packed macroblock records resident in HBM)",
            "config": {
                "workload": f"{args.width_mbs * 16}x{args.height_mbs * 16} ({args.width_mbs}x{args.height_mbs} MB) "
                            f"{args.profile}-profile IDR pictures, {'4x4 transform only' if args.profile == 'baseline' else '4x4+8x8 transform'}, "
                            f"{args.density} content (BASELINE.json configs[1])",
                "frames_per_gpu_per_step": F,
                "macroblocks_per_step": world * mbs_per_step,
                "stages": "dequant+IDCT+intra prediction+reconstruct -> planar YCbCr" + (" -> RGB" if want_rgb else ""),
                "parallelism": (("eight pictures per workgroup (8 lanes per picture)" if octl else
                                 "four pictures per workgroup (16 lanes per picture)" if quad else "one picture per workgroup")
                                + f", pictures sharded over {world} GPU(s), no collectives"),
                "bit_exact_vs_oracle": ok,
            },
            "kernel_ms": {recon_name: ms_recon, "ycbcr_to_rgb_kernel": ms_color},
            "host_frontend": None if host_rate is None else {
                "macroblocks_per_s_one_thread": host_rate, "stream_bytes_per_picture": stream_bytes / n_distinct,
                "note": "entropy decode is outside the timed region (inputs resident in HBM)"},
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None if traffic is None else traffic / 1e9,
                "traffic_unit": "GB per launch (rocprofv3 PMC, separate passes)",
                "traffic_source": traffic_src,
                "algorithmic_gb_per_launch": mbs_per_step * (bpm if dom_recon else BYTES_PER_MB_COLOR) / 1e9,
                "bytes_per_macroblock": bpm if dom_recon else BYTES_PER_MB_COLOR,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, rec, want_rgb, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if ok is False:
        raise SystemExit("bench: GPU output differs from the oracle")


if __name__ == "__main__":
    main()
