#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (per-dispatch counter rows) into per-kernel averages."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "recon_rows" in k:
            k = "recon_rows_kernel"
        elif "recon_quad" in k:
            k = "recon_quad_kernel"
        elif "recon_oct" in k:
            k = "recon_oct_kernel"
        elif "recon_pipe1" in k:
            k = "recon_pipe1_kernel"
        elif "recon_pipe" in k:
            k = "recon_pipe_kernel"
        elif "ycbcr_to_rgb" in k:
            k = "ycbcr_to_rgb_kernel"
        else:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
