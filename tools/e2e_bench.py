#!/usr/bin/env python3
"""End-to-end rate of the public API (not the bench.py metric): Annex-B file -> minivideo_open/parse/decode ->
picture files, i.e. including host entropy decoding (all host threads), PCIe both ways, the kernels and file
writes to a tmpfs directory.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--profile", default="baseline")
    ap.add_argument("--format", default="yuv420")
    ap.add_argument("--width-mbs", type=int, default=120)
    ap.add_argument("--height-mbs", type=int, default=68)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    from minivideo_amd import gen
    stream, _ = gen.make_stream(args.width_mbs, args.height_mbs, args.frames, seed=4242, profile=args.profile,
                                want_packed=False)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        path = os.path.join(d, "e2e.264")
        stream.tofile(path)
        env = dict(os.environ)
        if args.threads:
            env["MINIVIDEO_HOST_THREADS"] = str(args.threads)
        cli = os.path.join(ROOT, "minivideo_amd", "mini_thumbnailer")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "-i", path, "-f", args.format, "-n", str(min(args.frames, 999))], cwd=d, env=env,
                           capture_output=True, text=True)
        dt = time.perf_counter() - t0
        n_out = len([f for f in os.listdir(d) if f.startswith("e2e_") or f.startswith("e2e.")]) - 1
        ok = r.returncode == 0 and "did not succeed" not in r.stderr
    n = min(args.frames, 999)
    print(json.dumps({"what": "end-to-end CLI (stream file -> picture files on tmpfs)", "ok": ok, "pictures": n,
                      "files_written": n_out, "seconds": dt, "macroblocks_per_s": n * args.width_mbs * args.height_mbs / dt,
                      "fps": n / dt, "profile": args.profile, "format": args.format,
                      "stream_bytes": int(stream.size), "host_threads": args.threads or os.cpu_count()}))


if __name__ == "__main__":
    main()
