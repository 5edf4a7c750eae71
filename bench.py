#!/usr/bin/env python3
"""bench.py -- macroblocks/s of the MI355X-native H.264 intra (IDR) decode path.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1 runs one rank per GPU: either the caller starts the
ranks (`python -m torch.distributed.run ... bench.py --gpus N`), or, started bare, bench.py starts them itself before
anything touches a GPU; `--gpus` and WORLD_SIZE must agree, a mismatch is an error (never a silent 1-GPU run).

`value`: one "step" = one pass of the reconstruction hot path (dequantisation, IDCT, intra prediction, reconstruction,
fused RGB) over one batch of synthetic pictures whose packed macroblock records are already resident in HBM
(default: BASELINE.json configs[1], 2048 x 1080p Baseline per GPU, weak scaling; `--strong P` = configs[4], P pictures
split over the ranks).  Frames are independent: ranks share nothing on the data path; torch.distributed is only the
barrier and the max-over-ranks clock.

Rank 0 prints ONE JSON line with, besides the contract's keys,
  roofline      dominant kernel: algorithmic bytes / HIP-event duration against the 8 TB/s HBM peak;
  end_to_end    SURVEY 8(d)'s definition of the metric: stream bytes in host memory -> entropy decode on the host
                cores -> H2D -> kernels -> D2H -> planes + RGB in page-locked host memory (mvhp_engine_decode, the
                pipeline behind minivideo_decode), with the share of each stage and which one binds;
  cpu_baseline  (N = 1) the CPU restatement (oracle/, kind "port") on a bounded sample: reconstruction only (the span
                of `value`) and front end + reconstruction (the span of end_to_end), one thread and all host cores.
"""
import argparse
import json
import os
import subprocess
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
    ap.add_argument("--frames", type=int, default=2048, help="pictures per GPU per step (weak scaling)")
    ap.add_argument("--strong", type=int, default=0, metavar="P",
                    help="strong scaling: P pictures per step in total, split over the ranks (BASELINE.json configs[4]: 512)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pictures tiled to the batch")
    ap.add_argument("--width-mbs", type=int, default=120)
    ap.add_argument("--height-mbs", type=int, default=68)
    ap.add_argument("--profile", default="baseline", choices=["baseline", "high"])
    ap.add_argument("--density", default="dense", choices=["dense", "light"])
    ap.add_argument("--source", default="stream", choices=["stream", "records"],
                    help="stream: synthetic Annex-B stream -> host front end -> packed records (default); "
                         "records: random packed records drawn directly (minivideo_amd.synth)")
    ap.add_argument("--kinds", default="", metavar="P16,P8",
                    help="with --source records: P(Intra16x16), P(Intra8x8 | not Intra16x16) instead of the profile's mix "
                         "(content ablations: which macroblock kinds cost what; never a BASELINE.json configuration)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--layout", default="auto", choices=["auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"],
                    help="pictures per workgroup: rows = 1 (one wavefront per macroblock row), quad = 4, oct = 8; "
                         "wide / quad_wide = one picture / four pictures over several workgroups")
    ap.add_argument("--no-rgb", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="run the colour conversion as its own kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="budget of each of the four CPU baseline legs")
    ap.add_argument("--e2e-pictures", type=int, default=-1,
                    help="pictures of the end-to-end leg, in total over the ranks (-1: 2048 per rank, or P with --strong; 0: skip)")
    ap.add_argument("--e2e-repeats", type=int, default=3, help="timed calls of the end-to-end leg (the median is reported)")
    ap.add_argument("--e2e-batch", type=int, default=0, help="pictures per launch in the end-to-end leg (0: engine default)")
    ap.add_argument("--output-sets", type=int, default=3,
                    help="sets of ordinary output buffers the timed steps write in rotation (as the engine's three batch buffers per "
                         "context are); 1 = every step into the same buffers, as rounds 1-3 did")
    ap.add_argument("--placement-trials", type=int, default=1,
                    help="N = 1 only: re-time the launch this many times on the OTHER kind of buffers (placed when the timed steps "
                         "ran on ordinary allocations, and the other way round); reported beside value, never part of it.  A placed "
                         "set takes one allocation of up to 200 GB (seconds to get, seconds for the driver to clear afterwards)")
    ap.add_argument("--placed-buffers", action="store_true",
                    help="the timed steps run on buffers from mvhp_placed_alloc() (records / planes / RGB in three groups of the "
                         "device's memory regions, DESIGN.md 3); default since round 4: ordinary allocations")
    ap.add_argument("--ordinary-buffers", action="store_true", help="(the default since round 4; accepted for older scripts)")
    ap.add_argument("--host-threads", type=int, default=0, help="entropy threads per rank (0: host cores / ranks)")
    ap.add_argument("--e2e-placed", action="store_true",
                    help="the end-to-end leg's engine takes its batch buffers from a placed arena (MINIVIDEO_PLACED=1)")
    ap.add_argument("--engine-contexts", type=int, default=-1,
                    help="the single-process leg: ONE engine with this many contexts (one per device; more contexts than devices "
                         "share them = a rehearsal, never a result).  -1: --gpus when N > 1, 2 on one GPU (rehearsal); 0: skip")
    ap.add_argument("--cli-pictures", type=int, default=999,
                    help="pictures of the cold mini_thumbnailer run (a fresh process: stream file on tmpfs -> .yuv files); 0: skip")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks ourselves, before anything touches a GPU."""
    import socket
    import torch
    have = torch.cuda.device_count()   # (counting devices does not initialise the GPU on this image)
    if have < args.gpus and os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") != "1":   # (the rehearsal puts every rank on device 0)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) are visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def measured_traffic(args, fused, kernel, frames):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command
    (tools/profile_config.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950
    correction in MI355X_MICROARCH.md).  Only reported when the run uses a profiled configuration."""
    if not (args.density == "dense" and args.source == "stream" and fused and not args.waves and args.layout == "auto"):
        return None, None
    tag = {("baseline", 120, 68, 2048): "base1080", ("high", 120, 68, 2048): "high1080",
           ("high", 240, 135, 1024): "high2160", ("baseline", 120, 68, 512): "strong512",
           ("baseline", 120, 68, 64): "frames64"}.get((args.profile, args.width_mbs, args.height_mbs, frames))
    if tag is None:
        return None, None
    for rnd in ("r04q", "r04m", "r04h", "r04c", "r03b", "r03a", "r02f", "r02e", "r02d", "r02c", "r02b", "r02a"):
        name = f"{rnd}_{tag}_pmc_summary.json"
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            d = json.load(open(path)).get(kernel.split(" ")[0], {})
            if "hbm_read_bytes_corrected" in d and "hbm_write_bytes" in d:
                return d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"], "profiles/" + name
    return None, None


def other_frac(placement, algorithmic_bytes):
    """roofline fraction of the launch timed on the other kind of buffers (`placement`), or None"""
    if not placement:
        return None
    key = "ms_per_step_on_placed_buffers" if "ms_per_step_on_placed_buffers" in placement else "ms_per_step_on_ordinary_allocations"
    ms = [v for v in placement.get(key, []) if v]
    if not ms:
        return None
    return {"buffers": "mvhp_placed_alloc" if "placed" in key else "ordinary allocations",
            "ms_per_step": ms, "frac": [round(algorithmic_bytes / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for v in ms]}


def host_cores():
    """Cores this process may use: hardware threads cut down to the container's CPU quota (cgroup cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def repeat_stream(stream, n_distinct, total):
    """An Annex-B stream of `total` pictures: the parameter sets once, then the `n_distinct` IDR NAL units of `stream`
    over and over (every picture is entropy-decoded again; the 64 trailing zero bytes the ES index needs stay last)."""
    s = stream
    idx = np.flatnonzero((s[:-4] == 0) & (s[1:-3] == 0) & (s[2:-2] == 0) & (s[3:-1] == 1) & (s[4:] == 0x65))
    assert len(idx) == n_distinct, "unexpected start codes inside the synthetic stream"
    head, body, tail = s[:idx[0]], s[idx[0]:len(s) - 64], s[len(s) - 64:]
    reps, rem = divmod(total, n_distinct)
    parts = [head] + [body] * reps
    if rem:
        parts.append(s[idx[0]:idx[rem]])
    return np.concatenate(parts + [tail])


def cpu_baseline(params, stream, n_distinct, rec, want_rgb, budget_s):
    """The CPU restatement on the host cores (kind "port"): oracle/recon_ref.c for the reconstruction stages, the
    library's own host front end for the entropy stage it shares with the GPU path."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from minivideo_amd import lib
    from oracle import loader
    L = lib()
    cores = host_cores()
    h = C.c_void_p()
    have_stream = stream is not None and L.mvhp_stream_open(stream.ctypes.data, stream.size, C.byref(h)) == 1
    mbs = params.mbs

    def recon_one(k):
        loader.recon(params, rec[k % n_distinct], 1, want_rgb=want_rgb)

    def e2e_one(k):
        buf = np.empty(params.packed_bytes, np.uint8)
        if L.mvhp_stream_decode_packed(h, k % n_distinct, buf.ctypes.data, buf.size) != 1:
            raise RuntimeError("cpu_baseline: front end failed")
        loader.recon(params, buf, 1, want_rgb=want_rgb)

    def timed(fn, threads):
        fn(0)   # warm
        n, t0 = 0, time.perf_counter()
        if threads == 1:
            while time.perf_counter() - t0 < budget_s:
                fn(n)
                n += 1
        else:   # ctypes releases the GIL inside the C calls: the threads run on all cores
            with ThreadPoolExecutor(threads) as ex:
                while time.perf_counter() - t0 < budget_s:
                    list(ex.map(fn, range(n, n + 2 * threads)))
                    n += 2 * threads
        dt = time.perf_counter() - t0
        return n * mbs / dt, n, dt

    r1, n1, d1 = timed(recon_one, 1)
    rN, nN, dN = timed(recon_one, cores)
    out = {
        "value": r1,
        "unit": "macroblocks/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n1} of the benchmark's pictures ({n1 * mbs} macroblocks, {d1:.1f} s), oracle/recon_ref.c, one thread, "
                  f"the stages of the GPU step (packed records -> planes" + (" -> RGB)" if want_rgb else ")"),
        "cpu_model": cpu_model(),
        "host_cores": cores,
        "hardware_threads": os.cpu_count(),
        "all_cores": {"value": rN, "cores": cores, "sample": f"{nN} pictures, {dN:.1f} s, one picture per thread"},
    }
    if have_stream:
        e1, m1, t1 = timed(e2e_one, 1)
        eN, mN, tN = timed(e2e_one, cores)
        out["end_to_end"] = {
            "span": "stream bytes -> host front end (entropy decode) -> oracle/recon_ref.c -> planes" + (" + RGB" if want_rgb else ""),
            "value": e1, "unit": "macroblocks/s", "cores": 1, "sample": f"{m1} pictures, {t1:.1f} s",
            "all_cores": {"value": eN, "cores": cores, "sample": f"{mN} pictures, {tN:.1f} s"},
        }
        L.mvhp_stream_close(h)
    return out


def all_ranks_ok(ok, world, dist, dev):
    """MIN over the ranks of a per-rank verdict (a mismatch on a rank other than 0 must not pass silently)."""
    if not dist.is_initialized():
        return bool(ok)
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def end_to_end(args, params, stream, n_distinct, rec, want_rgb, total, rank, world, local_rank, dist, dev):
    """stream bytes in host memory -> planes (+ RGB) in page-locked host memory, through the pipeline
    behind minivideo_decode.  Every rank decodes its share of the pictures with its own engine on its own GPU."""
    import ctypes as C
    import torch
    from minivideo_amd import Engine, lib
    from minivideo_amd.dist import shard
    L = lib()
    cores = host_cores()
    threads = args.host_threads or max(1, cores // world)
    lo, hi = shard(total, rank, world)
    mine = hi - lo                           # every rank builds and decodes its own share of the pictures
    big = repeat_stream(stream, n_distinct, mine)
    h = C.c_void_p()
    if L.mvhp_stream_open(big.ctypes.data, big.size, C.byref(h)) != 1 or L.mvhp_stream_idr_count(h) != mine:
        raise SystemExit("bench: the end-to-end stream failed to parse")
    order = list(range(mine))
    # Device memory the PREVIOUS process released is wiped in the background (3 s for the 200-GB arena of a bench.py that has
    # just ended), and allocations wait for it: one allocation of what the engine will ask for, freed again, takes that wait
    # before the cold call is timed -- it is reported, and it is not the engine's.
    t_w = time.perf_counter()
    try:
        probe = torch.empty(int(40e9), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        del probe
        torch.cuda.empty_cache()
    except RuntimeError:
        pass
    wipe_wait_s = time.perf_counter() - t_w
    eng = Engine(contexts=1, host_threads=threads, batch_pictures=args.e2e_batch, first_device=local_rank, placed=args.e2e_placed)
    check = sorted({0, 1, len(order) // 2, len(order) - 1} & set(range(len(order))))
    kept = {}

    def sink(seq, idr, rc, err, p, yuv, rgb):
        if rc == 1 and seq in check:
            kept[seq] = (idr, yuv.copy(), rgb.copy() if rgb is not None else None)
        return 1 if rc == 1 else 0

    # cold call (reported, not the metric): the same job once, so that the engine owns its page-locked pools and its
    # device buffers at their working size; then the timed call on the same engine -- a service that decodes stream
    # after stream is in that state
    rc0, st0 = eng.decode(h, order, want_rgb=want_rgb)
    if dist.is_initialized():
        dist.barrier()
    # three timed calls, each bracketed like the kernel leg (barrier, synchronize, MAX over the ranks); the line reports the
    # MEDIAN -- one 0.5-s call varies by several per cent from call to call on this pool -- and lists all three
    calls, rc = [], 1
    for rep in range(max(1, args.e2e_repeats)):
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        rc_i, st_i = eng.decode(h, order, want_rgb=want_rgb, sink=sink)
        torch.cuda.synchronize(dev)
        w_i = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([w_i], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w_i = float(t.item())
        calls.append((w_i, st_i))
        rc = rc if rc_i == 1 and st_i["pictures_ok"] == len(order) else 0
    wall, st = sorted(calls, key=lambda c: c[0])[len(calls) // 2]   # (the stage figures below are the median call's)
    walls = [c[0] for c in calls]
    # the same job with RGB as the only output (what minivideo_decode asks for when it writes bmp / png / tga): a third of the
    # bytes cross the link on the way back
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    rc_r, st_r = eng.decode(h, order, want_rgb=3) if want_rgb else (1, None)
    torch.cuda.synchronize(dev)
    wall_r = time.perf_counter() - t1
    eng.close()
    L.mvhp_stream_close(h)
    ok = rc == 1 and rc_r == 1
    from oracle import loader   # every rank checks its own pictures (each rank decodes a stream of its own seed)
    for seq, (idr, yuv, rgb) in kept.items():
        ref, ref_rgb = loader.recon(params, rec[idr % n_distinct], 1, want_rgb=want_rgb)
        ok = ok and bool(np.array_equal(yuv, ref)) and (not want_rgb or bool(np.array_equal(rgb, ref_rgb)))
    ok = ok and len(kept) == len(check)
    ok = all_ranks_ok(ok, world, dist, dev)
    w = st["wall_s"]
    stages = {
        "entropy_decode_host": {"busy_s": st["entropy_busy_s"], "threads": st["host_threads"],
                                "share_of_wall": st["entropy_busy_s"] / max(1, st["host_threads"]) / w},
        "h2d": {"busy_s": st["h2d_s"], "share_of_wall": st["h2d_s"] / w, "GB/s": st["h2d_bytes"] / max(st["h2d_s"], 1e-9) / 1e9},
        "kernels": {"busy_s": st["kernel_s"], "share_of_wall": st["kernel_s"] / w, "launches": st["batches"],
                    "largest_batch": st["max_batch_pictures"],
                    "by_layout": dict(zip(["auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"], list(st["launches_by_layout"]) + list(st["launches_wide"])))},
        "d2h": {"busy_s": st["d2h_s"], "share_of_wall": st["d2h_s"] / w, "GB/s": st["d2h_bytes"] / max(st["d2h_s"], 1e-9) / 1e9},
    }
    bound = max(stages, key=lambda k: stages[k]["share_of_wall"])
    ent_share = stages["entropy_decode_host"]["share_of_wall"]
    return {
        "host": {
            "host_cores_visible": cores, "ranks_on_this_host": world, "host_cores_per_rank": cores / world,
            "entropy_threads_per_rank": threads,
            "note": "this leg is bound by entropy decoding on the host: N ranks share the host's cores, so its N-GPU curve is a "
                    "host-core curve (about 16 cores per GPU sustain 3e7 macroblocks/s per GPU on Baseline CAVLC, 1.4e7 on CABAC)"},
        "wall_not_entropy": {
            "share_of_wall": max(0.0, 1.0 - ent_share),
            "what": "the part of the call during which the entropy threads are not all busy: the engine's start (first chunk "
                    "page-locked and decoded before anything can be uploaded), the drain (upload + kernel + download of the "
                    "last batch, which nothing overlaps) and the threads that wait for a free chunk while a download is late"},
        "value": total * params.mbs / wall,
        "unit": "macroblocks/s",
        "span": "Annex-B bytes in host memory -> planes" + (" + RGB" if want_rgb else "") + " in page-locked host memory "
                "(mvhp_engine_decode: entropy threads || H2D || kernels || D2H)",
        "pictures": total,
        "n_gpus": world,
        "wall_s": wall, "wall_s_each": walls, "timed_calls": "median of %d calls on one engine, after its cold call" % len(walls),
        "outputs": "planes + RGB" if want_rgb else "planes",
        "rgb_only_rank0": None if st_r is None else {
            "value": mine * params.mbs / wall_r, "unit": "macroblocks/s", "wall_s": wall_r, "pictures": mine,
            "d2h_bytes_per_picture": st_r["d2h_bytes"] / max(1, mine),
            "d2h_share_of_wall": st_r["d2h_s"] / st_r["wall_s"],
            "entropy_share_of_wall": st_r["entropy_busy_s"] / max(1, st_r["host_threads"]) / st_r["wall_s"],
            "note": "this rank's share of the job, MVHP_OUT_RGB_ONLY: the planes stay on the device"},
        "buffers": "mvhp_placed_alloc" if st.get("placed_buffers") else "hipMalloc (ordinary allocations)",
        "cold_call_s": st0["wall_s"], "waited_for_the_previous_process_s": wipe_wait_s,
        "cold_call_pictures": st0["pictures_ok"],
        "cold_call": {"wall_s": st0["wall_s"], "first_picture_s": st0["first_picture_s"],
                      "page_locking_s": st0["host_alloc_s"], "page_locked_GB": st0["host_alloc_bytes"] / 1e9,
                      "device_alloc_s": st0["dev_alloc_s"], "device_alloc_GB": st0["dev_alloc_bytes"] / 1e9,
                      "first_launch_s": st0["first_launch_s"], "entropy_busy_s": st0["entropy_busy_s"],
                      "note": "the first call of a fresh engine (what one mini_thumbnailer run pays); the timed call reuses its pools"},
        "stream_bytes_per_picture": st["stream_bytes"] / max(1, len(order)),
        "stages_rank0": stages,
        "bound": bound,
        "bit_exact_vs_oracle": ok,
    }


def engine_multi_context(args, params, stream, n_distinct, rec, want_rgb, n_ctx, pictures, n_devices):
    """The in-library work queue north_star names: ONE process, ONE engine, one context per device, one shared pool
    of entropy threads, contexts claiming whole batches.  With more contexts than devices the contexts share devices: that
    exercises the code path and is labelled a rehearsal -- never a scaling result."""
    import ctypes as C
    from minivideo_amd import Engine, lib
    from oracle import loader
    L = lib()
    big = repeat_stream(stream, n_distinct, pictures)
    h = C.c_void_p()
    if L.mvhp_stream_open(big.ctypes.data, big.size, C.byref(h)) != 1 or L.mvhp_stream_idr_count(h) != pictures:
        raise SystemExit("bench: the multi-context stream failed to parse")
    order = list(range(pictures))
    check = sorted({0, pictures // 3, pictures // 2, pictures - 1})
    kept = {}

    def sink(seq, idr, rc, err, p, yuv, rgb):
        if rc == 1 and seq in check:
            kept[seq] = (idr, yuv.copy(), rgb.copy() if rgb is not None else None)
        return 1 if rc == 1 else 0

    eng = Engine(contexts=n_ctx, host_threads=args.host_threads, batch_pictures=args.e2e_batch, first_device=0)
    eng.decode(h, order, want_rgb=want_rgb)                       # cold call: pools and device buffers at working size
    t0 = time.perf_counter()
    rc, st = eng.decode(h, order, want_rgb=want_rgb, sink=sink)
    wall = time.perf_counter() - t0
    eng.close()
    L.mvhp_stream_close(h)
    ok = rc == 1 and st["pictures_ok"] == pictures and len(kept) == len(check)
    for seq, (idr, yuv, rgb) in kept.items():
        ref, ref_rgb = loader.recon(params, rec[idr % n_distinct], 1, want_rgb=want_rgb)
        ok = ok and bool(np.array_equal(yuv, ref)) and (not want_rgb or bool(np.array_equal(rgb, ref_rgb)))
    return {
        "what": "one process, one engine, %d contexts over %d device(s), one pool of %d entropy threads" % (n_ctx, n_devices, st["host_threads"]),
        "rehearsal_contexts_share_devices": n_ctx > n_devices,
        "value": pictures * params.mbs / wall, "unit": "macroblocks/s", "pictures": pictures, "wall_s": wall,
        "contexts": st["contexts"], "launches": st["batches"], "largest_batch": st["max_batch_pictures"],
        "by_layout": dict(zip(["auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"], list(st["launches_by_layout"]) + list(st["launches_wide"]))),
        "entropy_share_of_wall": st["entropy_busy_s"] / max(1, st["host_threads"]) / st["wall_s"],
        "bit_exact_vs_oracle": ok,
    }


def cli_cold(args, params, stream, n_distinct, rec):
    """mini_thumbnailer, the caller north_star names, as a user runs it: a fresh process per invocation (engine created,
    pools page-locked, code objects loaded, every time), stream file and .yuv files on tmpfs.  Reference span minivideo.c:255-303."""
    import shutil
    import tempfile
    from oracle import loader
    exe = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")
    n = min(args.cli_pictures, 999)                               # the CLI caps -n at 999 (main.cpp:186-196)
    if n <= 0 or not os.path.exists(exe):
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="mvbench_", dir=base)
    try:
        path = os.path.join(d, "clip.264")
        repeat_stream(stream, n_distinct, n).tofile(path)
        # three fresh processes (the files of one run are removed before the next); the MEDIAN run is reported: how long the HIP
        # runtime takes to come up in a new process varies between 0.05 and 0.25 s on this pool
        runs = []
        for rep in range(3):
            for f in os.listdir(d):
                if f.endswith(".yuv"):
                    os.unlink(os.path.join(d, f))
            t0 = time.perf_counter()
            r_i = subprocess.run([exe, "-i", path, "-f", "yuv420", "-n", str(n)], cwd=d, capture_output=True, text=True, timeout=600,
                                 env=dict(os.environ, MINIVIDEO_STATS="1"))
            runs.append((time.perf_counter() - t0, r_i))
            if r_i.returncode != 0:
                break
        wall, r = sorted(runs, key=lambda x: x[0])[len(runs) // 2] if all(x[1].returncode == 0 for x in runs) else runs[-1]
        files = [f for f in os.listdir(d) if f.endswith(".yuv")]
        ok = r.returncode == 0 and len(files) == n
        for k in (0, n // 2, n - 1):
            name = os.path.join(d, "clip_%d.yuv" % k if n > 1 else "clip.yuv")
            if ok and os.path.exists(name):
                ok = bool(np.array_equal(np.fromfile(name, np.uint8), loader.recon(params, rec[k % n_distinct], 1)[0]))
            else:
                ok = False
        stats = [l for l in r.stderr.splitlines() if l.startswith("[minivideo] decode:")]
        call = [l for l in r.stderr.splitlines() if l.startswith("[minivideo] decode call:") or l.startswith("[minivideo] parse call:")]
        # the CLI's DEFAULT output format (jpg falls back to png, export.c:652-658), once: the file writers' checksums over
        # 6.3 MB per picture are then the larger half of the CPU work
        for f in files:
            os.unlink(os.path.join(d, f))
        t0 = time.perf_counter()
        r_p = subprocess.run([exe, "-i", path, "-n", str(n)], cwd=d, capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, MINIVIDEO_STATS="1"))
        wall_p = time.perf_counter() - t0
        pngs = [f for f in os.listdir(d) if f.endswith(".png")]
        png = {"what": "the same with the default format (png)", "ok": r_p.returncode == 0 and len(pngs) == n,
               "value": n * params.mbs / wall_p, "unit": "macroblocks/s", "wall_s": wall_p,
               "bytes_written": sum(os.path.getsize(os.path.join(d, f)) for f in pngs)}
        return {"png": png, "what": "mini_thumbnailer -f yuv420 -n %d, a fresh process: stream file on tmpfs -> .yuv files on tmpfs" % n,
                "value": n * params.mbs / wall, "unit": "macroblocks/s", "pictures": n, "wall_s": wall,
                "wall_s_each": [x[0] for x in runs], "files_equal_oracle": ok, "library_stats": stats[-1] if stats else None,
                "library_calls": call or None}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box (never a result): every rank on device 0, control plane over gloo -- exercises the
    # sharding, the per-rank end-to-end leg and the rank-0 line without a second GPU
    rehearsal = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    # BENCH_RCCL_ONE_RANK=1 (tests/test_gpu_bench_rehearsal.py): an N = 1 run that makes every RCCL / gloo call of an N > 1 run
    # (process group on the device, side group, barriers, reductions) with a world of one -- all a one-GPU box can show of them
    use_dist = world > 1 or os.environ.get("BENCH_RCCL_ONE_RANK") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the reconstruction hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # a HOST-side group for waits during which another rank's work runs on this rank's device (rank 0's single-process engine
    # below): a barrier of the RCCL group is a kernel spinning on every GPU, under the collective watchdog
    host_pg = dist.new_group(backend="gloo") if (use_dist and not rehearsal) else None

    from minivideo_amd import HotPath
    from minivideo_amd.dist import shard

    want_rgb = not args.no_rgb
    if args.strong:
        lo, hi = shard(args.strong, rank, world)
        F = hi - lo
        if F <= 0:
            raise SystemExit("bench: --strong needs at least one picture per rank")
        total_frames = args.strong
    else:
        F = args.frames
        total_frames = F * world
    n_distinct = min(args.distinct, F)
    host_rate = None
    stream = None
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
        kinds = tuple(float(v) for v in args.kinds.split(",")) if args.kinds else None
        params, rec = synth_packed(args.width_mbs, args.height_mbs, n_distinct, seed=1000 + rank,
                                   profile=args.profile, density=args.density, kinds=kinds, qp_range=(24, 32))
        stream_bytes = None
    dev = torch.device("cuda", local_rank)
    # (the pipeline legs run FIRST: a process that has just released the kernel leg's 200-GB arena pays seconds for its next
    #  device allocations -- round 2's "cold call 6.2 s" was that, not the engine: tools/../profiles r03f)
    # ---- end to end (stream bytes -> host planes), every rank on its share ----
    # like the kernel leg: weak scaling by default (2048 pictures per rank), the fixed job with --strong
    e2e_total = args.e2e_pictures if args.e2e_pictures >= 0 else (args.strong or 2048 * world)
    e2e = None
    if e2e_total >= world and stream is not None:
        e2e = end_to_end(args, params, stream, n_distinct, rec, want_rgb, e2e_total, rank, world, local_rank, dist, dev)

    # ---- the single-process multi-context leg and the cold CLI run (rank 0; the other ranks wait at the barrier below) ----
    multi = cli = None
    if use_dist:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()           # rank 0's engine allocates on every device: leave it the room
        dist.barrier(group=host_pg)
    if rank == 0 and stream is not None:
        n_dev = torch.cuda.device_count() if not rehearsal else 1
        n_ctx = args.engine_contexts if args.engine_contexts >= 0 else (world if world > 1 else 2)
        # (two side legs: an environmental failure -- no room on the tmpfs for 6 GB of PNG files, a CLI run that hits its
        #  timeout -- is reported inside their block and does not take the headline line down with it)
        if n_ctx > 0 and e2e is not None:
            try:
                multi = engine_multi_context(args, params, stream, n_distinct, rec, want_rgb, n_ctx,
                                             (2048 * n_ctx) if n_ctx <= n_dev else 1024, n_dev)
            except Exception as ex:   # noqa: BLE001
                multi = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and args.cli_pictures > 0 and args.width_mbs * args.height_mbs <= 8160 and e2e is not None:
            try:
                cli = cli_cold(args, params, stream, n_distinct, rec)
            except Exception as ex:   # noqa: BLE001
                cli = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if use_dist:
        dist.barrier(group=host_pg)   # the other ranks start their kernel leg only when rank 0's engine has left their devices

    d_small = torch.from_numpy(rec.reshape(rec.shape[0], -1)).to(dev)
    reps = (F + d_small.shape[0] - 1) // d_small.shape[0]
    d_packed = d_small.repeat(reps, 1)[:F].contiguous()
    del d_small
    # The batch's buffers come from mvhp_placed_alloc(): records, planes and RGB each in a different group of the device's
    # memory regions (DESIGN.md 3 "Placement": the same launch takes 8.2 ms so placed and 9.6-10 ms with planes and RGB in
    # one group; ordinary allocations land in either case by chance -- they are timed below as `placement`, for comparison).
    # --ordinary-buffers, or an arena that cannot be had, falls back to ordinary allocations.
    import ctypes as C
    hipc = C.CDLL("libamdhip64.so")
    hipc.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    placed, buffers_info = None, {"allocator": "torch.empty (hipMalloc)"}
    if args.placed_buffers and not args.ordinary_buffers:
        from minivideo_amd import PlacedBuffers, MiniVideoError
        try:
            sizes = [d_packed.numel(), F * params.yuv_bytes] + ([F * params.rgb_bytes] if want_rgb else [])
            placed = PlacedBuffers(local_rank, sizes)
            buffers_info = {"allocator": "mvhp_placed_alloc", "groups_in_arena": placed.groups_found,
                            "group_of_records_planes_rgb": placed.groups, "seconds": round(placed.seconds, 2)}
        except MiniVideoError as ex:
            buffers_info["note"] = "mvhp_placed_alloc failed (%s)" % ex
    if placed:
        assert hipc.hipMemcpy(placed.ptrs[0], d_packed.data_ptr(), d_packed.numel(), 3) == 0
        p_packed, p_yuv, p_rgb = placed.ptrs[0], placed.ptrs[1], (placed.ptrs[2] if want_rgb else None)
        torch.cuda.synchronize(dev)
        del d_packed
        d_packed = d_yuv = d_rgb = None
        torch.cuda.empty_cache()
    else:
        d_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
        d_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev) if want_rgb else None
        p_packed, p_yuv, p_rgb = d_packed.data_ptr(), d_yuv.data_ptr(), (d_rgb.data_ptr() if want_rgb else None)
    # Output buffers in ROTATION (ordinary allocations only): consecutive steps write different sets, as the decode engine's
    # three batch buffers per context are used -- and so that `value` does not hang on where ONE set of allocations happened to
    # land in device memory (worth up to 25 % of this launch on some boxes).  --output-sets 1 = one set, as before.
    out_sets, rot_keep = [(p_yuv, p_rgb)], []
    if not placed:
        out_bytes = F * (params.yuv_bytes + (params.rgb_bytes if want_rgb else 0))
        n_sets = max(1, min(args.output_sets, int(100e9 // max(out_bytes, 1))))
        for _ in range(n_sets - 1):
            t_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
            t_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev) if want_rgb else None
            rot_keep += [t_yuv, t_rgb]
            out_sets.append((t_yuv.data_ptr(), t_rgb.data_ptr() if want_rgb else None))
    launch_no = [0]
    torch.cuda.synchronize(dev)   # inputs are resident before anything is launched on the bench stream
    hot = HotPath(local_rank)
    if args.waves:
        hot.set_waves_per_picture(args.waves)
    hot.set_layout(args.layout)
    fused = want_rgb and not args.no_fused
    hot.set_fused_color(fused)
    # a dedicated (non-null) stream: the C-ABI treats a NULL stream as "the context's own stream",
    # and the HIP events below must sit on the stream the kernels are launched on.
    stream_t = torch.cuda.Stream(device=dev)
    sp = stream_t.cuda_stream
    assert sp != 0
    def step(ev=None, out=None):
        if out is None:
            out = out_sets[launch_no[0] % len(out_sets)]
            launch_no[0] += 1
        o_yuv, o_rgb = out
        if ev is not None:
            ev[0].record(stream_t)
        hot.recon_stages_dev(params, p_packed, F, o_yuv, o_rgb, sp, 3 if fused else 1)
        if ev is not None:
            ev[1].record(stream_t)
        if want_rgb and not fused:
            hot.recon_stages_dev(params, p_packed, F, o_yuv, o_rgb, sp, 2)
        if ev is not None:
            ev[2].record(stream_t)

    def barrier():
        if use_dist:
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
    layout_name, waves_used = hot.last_launch()   # what the library chose (speed only)
    recon_name = {"rows": "recon_rows_kernel", "quad": "recon_quad_kernel", "oct": "recon_oct_kernel",
                  "wide": "recon_rows_kernel", "quad_wide": "recon_quad_kernel", "pipe": "recon_pipe_kernel", "pipe1": "recon_pipe1_kernel"}[layout_name]

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_recon = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    ms_color = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if (want_rgb and not fused) else 0.0

    mbs_per_step = F * params.mbs
    value = total_frames * params.mbs * args.steps / elapsed

    # spot-check pictures of the last step against the oracle (bit-exact) -- the checker, not the product
    # (every rank checks its own batch; the verdicts are reduced with MIN so that a mismatch on any rank fails the run)
    from oracle import loader
    ok = True
    p_yuv, p_rgb = out_sets[(launch_no[0] - 1) % len(out_sets)]   # what the last step wrote
    for f in sorted({0, 1, 2, 3, F // 2, F - 1} & set(range(F))):
        src = f % rec.shape[0]
        ref, ref_rgb = loader.recon(params, rec[src:src + 1], 1, want_rgb=want_rgb)
        got = np.empty(params.yuv_bytes, np.uint8)
        assert hipc.hipMemcpy(got.ctypes.data, p_yuv + f * params.yuv_bytes, params.yuv_bytes, 2) == 0
        ok = ok and bool(np.array_equal(got, ref))
        if want_rgb:
            got = np.empty(params.rgb_bytes, np.uint8)
            assert hipc.hipMemcpy(got.ctypes.data, p_rgb + f * params.rgb_bytes, params.rgb_bytes, 2) == 0
            ok = ok and bool(np.array_equal(got, ref_rgb))
    ok = all_ranks_ok(ok, world, dist, dev)
    # The same launch on the OTHER kind of buffers, for comparison (reported beside `value`, never part of it): --placement-trials N
    placement = None
    if world == 1 and args.placement_trials > 0:
        keep = (p_packed, p_yuv, p_rgb)
        hold, ms_other, ms_more_ordinary = [], [], []
        other_placed = None

        def time_current(out=None):
            out = out if out is not None else (p_yuv, p_rgb)
            torch.cuda.synchronize(dev)
            step(out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream_t)
            for _ in range(5):
                step(out=out)
            e1.record(stream_t)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / 5
        try:
            if placed:   # timed on placed buffers: ordinary allocations, several sets, the earlier ones kept so that the next land elsewhere
                t_packed = torch.empty(mbs_per_step * 800, dtype=torch.uint8, device=dev)
                assert hipc.hipMemcpy(t_packed.data_ptr(), p_packed, t_packed.numel(), 3) == 0
                hold.append(t_packed)
                for _ in range(args.placement_trials):
                    t_yuv = torch.empty(F * params.yuv_bytes, dtype=torch.uint8, device=dev)
                    t_rgb = torch.empty(F * params.rgb_bytes, dtype=torch.uint8, device=dev) if want_rgb else None
                    hold += [t_yuv, t_rgb]
                    p_packed, p_yuv, p_rgb = t_packed.data_ptr(), t_yuv.data_ptr(), (t_rgb.data_ptr() if want_rgb else None)
                    ms_other.append(time_current())
            else:        # timed on ordinary allocations: every set of the rotation on its own (how much this launch depends on where
                #              its buffers happen to lie, on THIS box), then one placed set
                for o in out_sets:
                    ms_more_ordinary.append(time_current(o))
                rot_keep.clear()
                t_yuv = t_rgb = None
                torch.cuda.empty_cache()
                from minivideo_amd import PlacedBuffers, MiniVideoError
                sizes = [mbs_per_step * 800, F * params.yuv_bytes] + ([F * params.rgb_bytes] if want_rgb else [])
                other_placed = PlacedBuffers(local_rank, sizes)
                assert hipc.hipMemcpy(other_placed.ptrs[0], p_packed, sizes[0], 3) == 0
                p_packed, p_yuv, p_rgb = other_placed.ptrs[0], other_placed.ptrs[1], (other_placed.ptrs[2] if want_rgb else None)
                for _ in range(args.placement_trials):
                    ms_other.append(time_current())
            hot.sync_check(sp)
        except (RuntimeError, Exception) as ex:   # noqa: BLE001 -- not enough memory beside the batch, or no arena to be had
            ms_other.append(None)
            placement_note = "%s: %s" % (type(ex).__name__, ex)
        else:
            placement_note = None
        p_packed, p_yuv, p_rgb = keep
        placement = {"ms_per_step_timed": round(ms_recon + ms_color, 3),
                     "timed_on": "mvhp_placed_alloc" if placed else "ordinary allocations",
                     ("ms_per_step_on_ordinary_allocations" if placed else "ms_per_step_on_placed_buffers"): [None if v is None else round(v, 3) for v in ms_other],
                     "ms_per_step_by_set_of_ordinary_allocations": [round(v, 3) for v in ms_more_ordinary] or None,
                     "placed_set_up_s": None if other_placed is None else round(other_placed.seconds, 2),
                     "note": placement_note or "5 launches per figure; the other kind of buffers is never part of `value`"}
        hold.clear()
        if other_placed is not None:
            other_placed.close()
        t_packed = t_yuv = t_rgb = None
    del d_packed, d_yuv, d_rgb
    hot.close()
    if placed:
        placed.close()
    torch.cuda.empty_cache()

    if rank == 0:
        dom_recon = ms_recon >= ms_color
        bpm = BYTES_PER_MB_FUSED if fused else BYTES_PER_MB_RECON
        if dom_recon:
            achieved = mbs_per_step * bpm / (ms_recon * 1e-3) / 1e9
            kname = recon_name + (" (fused RGB epilogue)" if fused else "")
        else:
            achieved = mbs_per_step * BYTES_PER_MB_COLOR / (ms_color * 1e-3) / 1e9
            kname = "ycbcr_to_rgb_kernel"
        traffic, traffic_src = measured_traffic(args, fused, kname, F)
        is_cfg = args.width_mbs == 120 and args.height_mbs == 68 and args.density == "dense" and not args.kinds
        if args.strong and is_cfg and args.profile == "baseline":
            cfg = f"BASELINE.json configs[4]: {args.strong} independent 1080p IDR pictures per step sharded over {world} GPU(s)"
        elif is_cfg and args.profile == "baseline":
            cfg = "BASELINE.json configs[1]"
        elif is_cfg and args.profile == "high":
            cfg = "BASELINE.json configs[2]"
        elif args.width_mbs == 240 and args.height_mbs == 135 and args.profile == "high":
            cfg = "BASELINE.json configs[3]"
        else:
            cfg = "not a BASELINE.json configuration"
        out = {
            "metric": "macroblocks/s on 1080p H.264 IDR frames; 1/2/4/8-GPU scaling",
            "value": value,
            "unit": "macroblocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "rehearsal_all_ranks_on_one_gpu": True if rehearsal else None,
            "rccl_calls_with_one_rank": True if (use_dist and world == 1) else None,
            "data": (f"synthetic ({n_distinct} distinct {args.density} pictures of a generated "
                     f"{'CAVLC' if args.profile == 'baseline' else 'CABAC'} Annex-B stream, entropy-decoded by the host front end, "
                     if args.source == "stream" else f"synthetic ({n_distinct} distinct random {args.density} pictures, ")
                    + "tiled to the batch; packed macroblock records resident in HBM)",
            "config": {
                "workload": f"{args.width_mbs * 16}x{args.height_mbs * 16} ({args.width_mbs}x{args.height_mbs} MB) "
                            f"{args.profile}-profile IDR pictures, {'4x4 transform only' if args.profile == 'baseline' else '4x4+8x8 transform'}, "
                            f"{args.density} content ({cfg})",
                "frames_per_gpu_per_step": F,
                "macroblocks_per_step": total_frames * params.mbs,
                "stages": "dequant+IDCT+intra prediction+reconstruct -> planar YCbCr" + (" -> RGB" if want_rgb else ""),
                "parallelism": ({"oct": "eight pictures per workgroup (8 lanes per picture)",
                                 "quad": "four pictures per workgroup (16 lanes per picture)",
                                 "rows": "one picture per workgroup",
                                 "wide": "one picture over several workgroups (bands of macroblock rows, seams through global memory)",
                                 "quad_wide": "four pictures (16 lanes each) over several workgroups (bands of macroblock rows, "
                                              "seams through global memory)",
                                 "pipe": "four pictures (16 lanes each) over several workgroups (bands of four macroblock rows, seams "
                                         "through global memory), three wavefronts per row: residuals / luma / chroma + write-out",
                                 "pipe1": "one picture over several workgroups (bands of four macroblock rows, seams through global memory), "
                                          "three wavefronts per row: residuals / luma / chroma + write-out"}[layout_name]
                                + f", {waves_used} wavefronts per workgroup, pictures sharded over {world} GPU(s), no collectives"),
                "bit_exact_vs_oracle": ok,
            },
            "kernel_ms": {recon_name: ms_recon, "ycbcr_to_rgb_kernel": ms_color},
            "buffers": dict(buffers_info, output_sets_in_rotation=len(out_sets)),
            "placement": placement,
            "host_frontend": None if host_rate is None else {
                "macroblocks_per_s_one_thread": host_rate, "stream_bytes_per_picture": stream_bytes / n_distinct,
                "note": "entropy decode is outside the timed region of `value` (inputs resident in HBM); it is inside end_to_end"},
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
                # the same launch on the other kind of buffers (placement: where a batch's buffers lie in device memory is
                # worth up to 25 % of a launch on some boxes, nothing on others -- DESIGN.md 7); never part of `frac`
                "frac_on_the_other_buffers": other_frac(placement, mbs_per_step * (bpm if dom_recon else BYTES_PER_MB_COLOR)),
            },
            "end_to_end": e2e,
            "engine_multi_context": multi,
            "cli": cli,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(params, stream, n_distinct, rec, want_rgb, args.cpu_seconds)
            out["cpu_baseline"] = cb
            if e2e and "end_to_end" in cb:
                # not `vs_baseline` (BASELINE.md holds no published number for this metric): the measured ratios
                e2e["vs_cpu_port_one_thread"] = e2e["value"] / cb["end_to_end"]["value"]
                e2e["vs_cpu_port_all_cores"] = e2e["value"] / cb["end_to_end"]["all_cores"]["value"]
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # (a leg that could not run -- its block holds "error" -- has compared nothing; one that ran and differs fails the bench)
    if rank == 0 and ((multi is not None and "error" not in multi and not multi["bit_exact_vs_oracle"]) or
                      (cli is not None and "error" not in cli and not cli["files_equal_oracle"])):
        raise SystemExit("bench: the multi-context / CLI leg differs from the oracle")
    if ok is False or (e2e is not None and not e2e["bit_exact_vs_oracle"]):
        raise SystemExit("bench: GPU output differs from the oracle")


if __name__ == "__main__":
    main()
