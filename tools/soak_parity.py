"""GPU soak: random picture sizes, batch sizes, profiles, QP ranges, layouts and wave counts against the CPU oracle
(bit-exact).  usage: python tools/soak_parity.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
h = HotPath(0)
t0 = time.time()
n_cases = n_mb = 0
t_print = t0
while time.time() - t0 < budget:
    W = int(rng.integers(1, 40)); H = int(rng.integers(1, 40)); n = int(rng.integers(1, 14))
    if rng.random() < 0.1:
        W, H, n = 120, 68, int(rng.integers(1, 6))
    prof = ["baseline", "high"][int(rng.integers(0, 2))]
    dens = ["dense", "light"][int(rng.integers(0, 2))]
    lo = int(rng.integers(0, 40)); hi = int(rng.integers(lo, 52))
    layout = ["rows", "quad", "oct"][int(rng.integers(0, 3))]
    waves = [0, 4, 6, 8, 12, 16][int(rng.integers(0, 6))]
    rgb = bool(rng.integers(0, 2))
    kw = dict(profile=prof, density=dens, qp_range=(lo, hi), cqp_offsets=(int(rng.integers(-12, 13)), int(rng.integers(-12, 13))))
    if rng.random() < 0.2:
        kw["illegal_modes"] = True
    if lo <= 36 <= hi and rng.random() < 0.5:
        kw["allow_qp36_i16"] = True
    params, rec = synth_packed(W, H, n, seed=int(rng.integers(0, 1 << 30)), **kw)
    h.set_layout(layout); h.set_waves_per_picture(waves)
    g, gr = h.recon_host(params, rec, n, want_rgb=rgb)
    o, orr = loader.recon(params, rec, n, want_rgb=rgb)
    if not np.array_equal(g, o) or (rgb and not np.array_equal(gr, orr)):
        print("MISMATCH", W, H, n, prof, dens, (lo, hi), layout, waves, rgb, kw, flush=True)
        sys.exit(1)
    n_cases += 1; n_mb += W * H * n
    if time.time() - t_print > 30:   # progress line (a silent GPU job is taken to be hung)
        t_print = time.time()
        print("... %d cases, %d macroblocks, %.0f s" % (n_cases, n_mb, time.time() - t0), flush=True)
print("soak ok: %d cases, %d macroblocks, %.0f s" % (n_cases, n_mb, time.time() - t0))
