"""GPU soak: random picture sizes, batch sizes, profiles, QP ranges, layouts and wave counts against the CPU oracle
(bit-exact).  usage: python tools/soak_parity.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
n_spec = 0
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
    layout = ["rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1", "wide", "quad_wide", "pipe", "pipe1", "auto"][int(rng.integers(0, 12))]
    waves = [0, 1, 2, 4, 6, 8, 12, 16][int(rng.integers(0, 8))]   # (waves per workgroup / rows per band; a form takes what it is built for)
    if rng.random() < 0.05:   # round 4: enough pictures for several groups per band and a band-major ticket order that matters
        n = int(rng.integers(14, 80))
        W, H = min(W, 24), min(H, 40)
    rgb = bool(rng.integers(0, 2))
    kw = dict(profile=prof, density=dens, qp_range=(lo, hi), cqp_offsets=(int(rng.integers(-12, 13)), int(rng.integers(-12, 13))))
    if rng.random() < 0.2:
        kw["illegal_modes"] = True
    if lo <= 36 <= hi and rng.random() < 0.5:
        kw["allow_qp36_i16"] = True
    if rng.random() < 0.15:   # round 3: MVHP_STREAM_SPEC streams -- several slices, I_PCM, scaling lists (generator -> front end -> records)
        from minivideo_amd import gen
        from tests.util import Stream
        sprof = ["baseline", "main", "high", "high_cavlc"][int(rng.integers(0, 4))]
        W, H, n = min(W, 24), min(H, 16), min(n, 4)
        stream, packed, _ = gen.make_stream_ex(W, H, n, seed=int(rng.integers(0, 1 << 30)), profile=sprof, slices=int(rng.integers(1, 6)),
                                               pcm_permille=int(rng.choice([0, 40, 300])), scaling=int(rng.integers(0, 4)) if sprof.startswith("high") else 0,
                                               qp_range=(lo, hi))
        with Stream(stream, spec=True) as st:
            params = st.params(0)
            rec = np.stack([st.packed(k)[1] for k in range(n)])
        if not np.array_equal(rec.reshape(packed.shape), packed):
            print("FRONT END != GENERATOR", W, H, n, sprof, flush=True)
            sys.exit(1)
        n_spec = globals().get("n_spec", 0) + 1
    elif rng.random() < 0.2:   # reference-mode streams through the host front end (CAVLC and CABAC), then the compact path too
        from minivideo_amd import gen
        from tests.compact import decode_compact, expand as expand_compact
        from tests.util import Stream
        sprof = ["baseline", "main", "high", "high_cavlc"][int(rng.integers(0, 4))]
        W, H, n = min(W, 30), min(H, 20), min(n, 4)
        stream, packed = gen.make_stream(W, H, n, seed=int(rng.integers(0, 1 << 30)), profile=sprof, dense=(dens == "dense"), qp_range=(lo, hi), max_level=int(rng.choice([32, 300, 2000])))
        with Stream(stream) as st:
            params = st.params(0)
            rec = np.stack([st.packed(k)[1] for k in range(n)])
            for k in range(n):
                rc2, used, buf = decode_compact(st, k)
                if rc2 != 1 or not np.array_equal(expand_compact(buf, W * H).reshape(-1), rec[k].reshape(-1)):
                    print("COMPACT != PACKED", W, H, n, sprof, k, flush=True)
                    sys.exit(1)
        if not np.array_equal(rec.reshape(packed.shape), packed):
            print("FRONT END != GENERATOR", W, H, n, sprof, flush=True)
            sys.exit(1)
        n_ref = globals().get("n_ref", 0) + 1
    else:
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
print("soak ok: %d cases (%d of them spec-mode streams, %d reference-mode streams through the front end), %d macroblocks, %.0f s" % (n_cases, globals().get("n_spec", 0), globals().get("n_ref", 0), n_mb, time.time() - t0))
