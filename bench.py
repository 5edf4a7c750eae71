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
BYTES_PER_MB_RECON = 800 + 384  # packed record in + planar YCbCr out
BYTES_PER_MB_COLOR = 384 + 768  # colour kernel: YCbCr in + RGB out (B_rgb - B_yuv = 768 written)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=512, help="pictures per GPU per step")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pictures tiled to --frames")
    ap.add_argument("--width-mbs", type=int, default=120)
    ap.add_argument("--height-mbs", type=int, default=68)
    ap.add_argument("--profile", default="baseline", choices=["baseline", "high"])
    ap.add_argument("--density", default="dense", choices=["dense", "light"])
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--no-rgb", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


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
    from minivideo_amd.synth import synth_packed

    want_rgb = not args.no_rgb
    F = args.frames
    params, rec = synth_packed(args.width_mbs, args.height_mbs, min(args.distinct, F), seed=1000 + rank,
                               profile=args.profile, density=args.density)
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
    # a dedicated (non-null) stream: the C-ABI treats a NULL stream as "the context's own stream",
    # and the HIP events below must sit on the stream the kernels are launched on.
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    assert sp != 0
    rgb_ptr = d_rgb.data_ptr() if want_rgb else None

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        hot.recon_stages_dev(params, d_packed.data_ptr(), F, d_yuv.data_ptr(), rgb_ptr, sp, 1)
        if ev is not None:
            ev[1].record(stream)
        if want_rgb:
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
    ms_color = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if want_rgb else 0.0

    mbs_per_step = F * params.mbs
    value = world * mbs_per_step * args.steps / elapsed

    # spot-check one picture of the last step against the oracle (bit-exact) -- the checker, not the product
    ok = None
    if rank == 0:
        from oracle import loader
        yuv0 = d_yuv[: params.yuv_bytes].cpu().numpy()
        ref, _ = loader.recon(params, rec[:1], 1, want_rgb=False)
        ok = bool(np.array_equal(yuv0, ref))

    if rank == 0:
        dom_recon = ms_recon >= ms_color
        if dom_recon:
            achieved = mbs_per_step * BYTES_PER_MB_RECON / (ms_recon * 1e-3) / 1e9
            kname = "recon_rows_kernel"
        else:
            achieved = mbs_per_step * BYTES_PER_MB_COLOR / (ms_color * 1e-3) / 1e9
            kname = "ycbcr_to_rgb_kernel"
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
            "data": f"synthetic ({min(args.distinct, F)} distinct random {args.density} pictures tiled to the batch, "
                    "packed macroblock records resident in HBM)",
            "config": {
                "workload": f"{args.width_mbs * 16}x{args.height_mbs * 16} ({args.width_mbs}x{args.height_mbs} MB) "
                            f"{args.profile}-profile IDR pictures, {'4x4 transform only' if args.profile == 'baseline' else '4x4+8x8 transform'}, "
                            f"{args.density} content (BASELINE.json configs[1])",
                "frames_per_gpu_per_step": F,
                "macroblocks_per_step": world * mbs_per_step,
                "stages": "dequant+IDCT+intra prediction+reconstruct -> planar YCbCr" + (" -> RGB" if want_rgb else ""),
                "parallelism": f"frame-per-workgroup, {world} GPU(s), no collectives",
                "bit_exact_vs_oracle": ok,
            },
            "kernel_ms": {"recon_rows_kernel": ms_recon, "ycbcr_to_rgb_kernel": ms_color},
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "bytes_per_macroblock": BYTES_PER_MB_RECON if dom_recon else BYTES_PER_MB_COLOR,
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
