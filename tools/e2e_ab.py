#!/usr/bin/env python3
"""A/B of engine settings on the end-to-end span (stream bytes in host memory -> planes + RGB in page-locked memory), on
the GPU box: one stream, one process, the configurations run in turn for several rounds, median / min / max of the
wall time per configuration.  A single call varies by +-7 % from run to run on this pool; decisions need this.
usage: e2e_ab.py [--pictures 2048] [--rounds 7] [--profile baseline] [--rgb 1] name:key=value,key=value ...
   keys: batch (Engine batch_pictures), ctx (contexts; more than the devices = several contexts per device), any MINIVIDEO_* environment variable (e.g. MINIVIDEO_IN_CHUNKS=8)"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--profile", default="baseline")
    ap.add_argument("--rgb", type=int, default=1, help="0 planes, 1 planes + RGB, 3 RGB only")
    ap.add_argument("--width-mbs", type=int, default=120)
    ap.add_argument("--height-mbs", type=int, default=68)
    ap.add_argument("--pysink", type=int, default=0, help="1: a Python sink per picture (as bench.py's timed call has)")
    ap.add_argument("--torch", type=int, default=0, help="1: import torch and initialise its device context first (as bench.py does)")
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    if args.torch:
        import torch
        torch.cuda.synchronize(torch.device("cuda", 0))
        if args.torch > 1:
            x = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")   # (the caching allocator owns a block)
            torch.cuda.synchronize()
    import bench
    from minivideo_amd import Engine, gen, lib
    L = lib()
    n_distinct = 16
    stream, _ = gen.make_stream(args.width_mbs, args.height_mbs, n_distinct, seed=1000, profile=args.profile, dense=True, want_packed=False)
    big = bench.repeat_stream(stream, n_distinct, args.pictures)
    h = C.c_void_p()
    if L.mvhp_stream_open(big.ctypes.data, big.size, C.byref(h)) != 1:
        raise SystemExit("stream failed to parse")
    order = list(range(args.pictures))
    cfgs = []
    for spec in args.configs:
        name, _, kv = spec.partition(":")
        d = dict(x.split("=") for x in kv.split(",") if x)
        cfgs.append((name, d))
    engines, walls, shares = {}, {n: [] for n, _ in cfgs}, {n: [] for n, _ in cfgs}

    def apply_env(d):
        for k in list(os.environ):
            if k.startswith("MINIVIDEO_") and k not in ("MINIVIDEO_LIB",):
                del os.environ[k]
        for k, v in d.items():
            if k.startswith("MINIVIDEO_"):
                os.environ[k] = v

    for name, d in cfgs:
        apply_env(d)
        engines[name] = Engine(contexts=int(d.get("ctx", 1)), batch_pictures=int(d.get("batch", 0)))
        engines[name].decode(h, order, want_rgb=args.rgb)          # pools at working size
    for r in range(args.rounds):
        for name, d in (cfgs if r % 2 == 0 else cfgs[::-1]):
            apply_env(d)
            t0 = time.perf_counter()
            rc, st = engines[name].decode(h, order, want_rgb=args.rgb, sink=(lambda seq, idr, rc, err, p, yuv, rgb: 1 if rc == 1 else 0) if args.pysink else None)
            w = time.perf_counter() - t0
            assert rc == 1 and st["pictures_ok"] == args.pictures
            walls[name].append(w)
            shares[name].append((st["entropy_busy_s"] / st["host_threads"] / w, st["h2d_s"] / w, st["d2h_s"] / w, st["batches"]))
    mbs = args.pictures * args.width_mbs * args.height_mbs
    for name, _ in cfgs:
        w = walls[name]
        s = shares[name]
        print("%-22s median %.4f s = %.3e MB/s (min %.4f max %.4f)  entropy %.2f h2d %.2f d2h %.2f launches %d" % (
            name, statistics.median(w), mbs / statistics.median(w), min(w), max(w), statistics.median(x[0] for x in s),
            statistics.median(x[1] for x in s), statistics.median(x[2] for x in s), s[0][3]), flush=True)
    print(json.dumps({"pictures": args.pictures, "rounds": args.rounds, "profile": args.profile, "rgb": args.rgb,
                      "walls": walls}))


if __name__ == "__main__":
    main()
