#!/usr/bin/env python3
"""Instruction counts per source section of a batch kernel's macroblock loop, from a -DMVHP_MARKS build's ISA text
(the MVHP_MARK(name) comments of recon_oct.hip).  A measurement aid: the markers are volatile asm and perturb the
schedule a little; counts are static (per execution of each section's code, both sides of divergent branches).
usage: isa_sections.py file.s [kernel-name-substring]"""
import re
import sys
from collections import Counter, OrderedDict

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "recon_oct_kernelILi8ELb1"
L = open(path).read().split("\n")
s = next(i for i, l in enumerate(L) if re.match(r"^_ZN4mvhp\S*" + re.escape(want) + r"\S*:", l))
e = next(i for i in range(s, len(L)) if "s_endpgm" in L[i])
sec, counts = "prologue", OrderedDict()
for l in L[s:e]:
    m = re.search(r"; MARK (\w+)", l)
    if m:
        sec = m.group(1)
        continue
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        k = "valu"
    elif op.startswith("s_waitcnt"):
        k = "waitcnt"
    elif op.startswith("s_nop"):
        k = "nop"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        k = "branch"
    elif op.startswith("s_"):
        k = "salu"
    elif op.startswith("ds_"):
        k = "lds"
    elif op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_"):
        k = "vmem"
    else:
        k = "other"
    counts.setdefault(sec, Counter())[k] += 1
tot = Counter()
print("%-14s %6s %6s %5s %5s %7s %6s %4s" % ("section", "valu", "salu", "lds", "vmem", "waitcnt", "branch", "nop"))
for sec, c in counts.items():
    tot.update(c)
    print("%-14s %6d %6d %5d %5d %7d %6d %4d" % (sec, c["valu"], c["salu"], c["lds"], c["vmem"], c["waitcnt"], c["branch"], c["nop"]))
print("%-14s %6d %6d %5d %5d %7d %6d %4d" % ("total", tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["waitcnt"], tot["branch"], tot["nop"]))
