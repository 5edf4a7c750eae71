"""GPU sweep: pictures much wider than tall, and much taller than wide (121 ... 510 macroblocks per row / rows per picture: the widths at which the batch kernels' line buffers stop
fitting LDS and pick_layout / pick_waves fall back) on every forced layout and on the automatic choice, against the CPU oracle.
usage: python tools/soak_wide_pictures.py"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from minivideo_amd import HotPath
from minivideo_amd.synth import synth_packed
from oracle import loader

h = HotPath(0)
t0 = time.time()
cases = 0
took = {}
SHAPES = [(W, H) for W in (121, 160, 161, 200, 240, 255, 300, 400, 510) for H in (3, 9)] + [(W, H) for W in (1, 2, 7) for H in (121, 300, 510)]
for (W, H) in SHAPES:
    if True:
        for prof in ("baseline", "high"):
            params, rec = synth_packed(W, H, 5, seed=W + H, profile=prof, density="dense")
            ref = [loader.recon(params, rec[k], 1, want_rgb=True) for k in range(5)]
            for n in (1, 5, 40):
                recn = np.concatenate([rec] * ((n + 4) // 5))[:n]
                for layout in ("auto", "rows", "quad", "oct", "wide", "quad_wide", "pipe", "pipe1"):
                    for waves in (0, 4, 8):
                        h.set_layout(layout); h.set_waves_per_picture(waves)
                        g, gr = h.recon_host(params, recn, n, want_rgb=True)
                        g = g.reshape(n, -1); gr = gr.reshape(n, -1)
                        for f in range(n):
                            if not (np.array_equal(g[f], ref[f % 5][0].reshape(-1)) and np.array_equal(gr[f], ref[f % 5][1].reshape(-1))):
                                print("MISMATCH", W, H, prof, n, layout, waves, "picture", f, h.last_launch(), flush=True)
                                sys.exit(1)
                        took.setdefault((layout, h.last_launch()[0]), 0)
                        took[(layout, h.last_launch()[0])] += 1
                        cases += 1
    print("... %d x %d done, %d cases, %.0f s" % (W, H, cases, time.time() - t0), flush=True)
print("wide pictures ok: %d cases, %.0f s; asked -> ran: %s" % (cases, time.time() - t0, sorted(took.items())))
