#!/usr/bin/env python3
"""Print a gpurun_out/<tag>_pmc_sq_breakdown.json as per-macroblock / per-wave-life figures.  usage: sq_print.py file.json macroblocks ms"""
import json, sys
d = json.load(open(sys.argv[1])); mbs = float(sys.argv[2]); ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0
k = [x for x in d if 'recon' in x][0]; s = d[k]; wc = s["SQ_WAVE_CYCLES"]
print(k)
print("  per MB: VALU %.1f SALU %.1f LDS %.1f branch %.1f" % (s["SQ_INSTS_VALU"]/mbs, s["SQ_INSTS_SALU"]/mbs, s["SQ_INSTS_LDS"]/mbs, s.get("SQ_INSTS_BRANCH",0)/mbs))
print("  wave life: active %.3f (valu %.3f sca %.3f lds %.3f misc %.3f) wait_any %.3f wait_inst %.3f (lds %.3f)" % (
    s["SQ_ACTIVE_INST_ANY"]/wc, s["SQ_ACTIVE_INST_VALU"]/wc, s["SQ_ACTIVE_INST_SCA"]/wc, s["SQ_ACTIVE_INST_LDS"]/wc, s["SQ_ACTIVE_INST_MISC"]/wc,
    s["SQ_WAIT_ANY"]/wc, s["SQ_WAIT_INST_ANY"]/wc, s["SQ_WAIT_INST_LDS"]/wc))
print("  wave cycles (quad) per MB %.1f; LDS idx_active per MB %.1f, bank conflict %.3f of it; valu thread-cycles/inst %.1f" % (
    wc/mbs, s["SQ_LDS_IDX_ACTIVE"]/mbs, s["SQ_LDS_BANK_CONFLICT"]/s["SQ_LDS_IDX_ACTIVE"], s["SQ_THREAD_CYCLES_VALU"]/s["SQ_INSTS_VALU"]))
