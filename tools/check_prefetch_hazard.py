#!/usr/bin/env python3
"""Static check of the quad kernel's ISA.  The record prefetch lives in v100-v123, registers that only inline assembly
names (the compiler is limited to v0-v99 by amdgpu_num_vgpr): (a) no compiler-generated instruction may touch them;
(b) the guarded wait in front of their use is `s_waitcnt vmcnt(N)` with N = the number of stores a full-strip step
issues behind the prefetch, so exactly N asm stores must sit in the macroblock loop; (c) no scratch access (register
spill = a vector-memory operation the counted waits do not know about) may sit in that loop; (d) every inline-assembly
block that issues vector-memory instructions pads the hazards inline assembly hides from the compiler: >= 5 wait states
(`s_nop 4`) in front of its first vector-memory instruction -- the scalar base may have been written by a `v_readlane`
right in front of the block, and a vector-memory read of a VALU-written SGPR needs that distance (the round-1 GPU memory
fault) -- and >= 2 wait states (`s_nop 1`) behind a 128-bit store before its data registers may be rewritten.
The oct kernel's prefetch registers are v216-v247.
Violations raise HazardError (never a bare assert: the check must survive `python -O`).
usage: check_prefetch_hazard.py file.s"""
import os
import re
import sys


class HazardError(Exception):
    pass


def require(cond, *info):
    if not cond:
        raise HazardError(info)


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


# kernel family -> (prefetch registers, asm loads per prefetch block, stores of a full strip without / with RGB)
FAMILIES = {
    "recon_quad_kernel": (set(range(100, 124)), 6, (8, 20)),
    "recon_oct_kernel": (set(range(216, 248)), 8, (16, 40)),
}


def _in_asm(L, i):
    for k in range(i, -1, -1):
        if '#ASMEND' in L[k]:
            return False
        if '#ASMSTART' in L[k]:
            return True
    return False


def check(L, name):
    """L = the lines of one kernel."""
    fam = next(v for k, v in FAMILIES.items() if k in name)
    PREFETCH, n_block, strip_counts = fam
    # measurement builds that leave stores out (tools/build_variant.sh -DMVHP_ABL_*) name their own counts
    if os.environ.get("MVHP_CHECK_STRIP_COUNTS"):
        strip_counts = tuple(int(v) for v in os.environ["MVHP_CHECK_STRIP_COUNTS"].split(","))
    first = "v[%d:%d]" % (min(PREFETCH), min(PREFETCH) + 1)
    in_asm, touched, asm_loads, asm_stores_at, seam_stores = False, [], [], [], []
    wide = "recon_quad_kernel" in name and name.endswith("Lb1EEEvNS_9ReconArgsE")   # recon_quad_kernel<NW, RGB, true>
    for i, raw in enumerate(L):
        if '#ASMSTART' in raw:
            in_asm = True
            continue
        if '#ASMEND' in raw:
            in_asm = False
            continue
        l = raw.split(';')[0]
        if not l.strip() or l.strip().startswith('.'):
            continue
        if in_asm:
            if re.match(r'\s+global_load_dwordx4', l):
                asm_loads.append(i)
                require(regs(l.split(',')[0]) <= PREFETCH, name, l)
            elif re.match(r'\s+global_store', l):
                require(not (regs(l) & PREFETCH), name, l)
                if re.search(r'\bsc1\b', l):
                    seam_stores.append(i)    # wide instantiations: the granules for the band below (counted by the waits, below)
                else:
                    asm_stores_at.append(i)
        elif regs(l) & PREFETCH:
            touched.append((i, l))
    require(len(asm_loads) == 2 * n_block, name, len(asm_loads))      # one block in the prologue, one in the loop
    # wide instantiations of the quad kernel (recon_quad_kernel<NW, RGB, true>): the seam granule a lane has asked for lands in
    # v[124:125], requested two steps before its use -- registers the compiler must never name at all
    require(len(seam_stores) == (1 if wide else 0), name, 'asm seam stores', seam_stores)
    if wide:
        seam = [(i, l) for i, l in enumerate(L) if re.match(r'\s+global_load_dwordx2 v\[124:125\].* sc1', l.split(';')[0])]
        require(len(seam) >= 2, name, 'asm seam loads expected (blocking form and request)', seam)
        in_a, bad = False, []
        for i, raw in enumerate(L):
            if '#ASMSTART' in raw:
                in_a = True
            elif '#ASMEND' in raw:
                in_a = False
            elif not in_a and regs(raw.split(';')[0]) & {124, 125}:
                bad.append((i, raw))
        require(not bad, name, 'compiler code names the seam registers v124 / v125', bad[:5])
    # (d) hazard padding inside every asm block that touches vector memory
    blk = None
    for i, raw in enumerate(L):
        if '#ASMSTART' in raw:
            blk = []
        elif '#ASMEND' in raw and blk is not None:
            vm = [k for k, t in enumerate(blk) if re.match(r'\s*global_(load|store)', t)]
            if vm:
                pad = [int(m.group(1)) for t in blk[:vm[0]] for m in [re.match(r'\s*s_nop\s+(\d+)', t)] if m]
                require(pad and max(pad) >= 4, name, 'vector-memory asm without 5 wait states in front (v_readlane -> VMEM base hazard)', i)
                for k in vm:
                    if re.match(r'\s*global_store_dwordx4', blk[k]):
                        nxt = blk[k + 1] if k + 1 < len(blk) else ''
                        require(re.match(r'\s*s_nop\s+[1-9]', nxt), name, '128-bit asm store without 2 wait states behind it', i)
            blk = None
        elif blk is not None:
            t = raw.split(';')[0]
            if t.strip():
                blk.append(t)
    # the guarded waits: asm blocks that start with s_waitcnt vmcnt(N) and then move the prefetch registers out
    waits = [i for i, l in enumerate(L) if re.search(r's_waitcnt vmcnt\(\d+\)', l) and first in L[i + 1]]
    counts = sorted(int(re.search(r'vmcnt\((\d+)\)', L[w]).group(1)) for w in waits)
    if wide:   # a wave that feeds a seam has one more store behind its prefetch: waits for N + 1
        require(len(waits) == 4, name, waits)
        require(counts[0] == 0 and counts[1] == 1 and counts[2] in strip_counts and counts[3] == counts[2] + 1, name, counts)
        n_expect = counts[2]
    else:
        require(len(waits) == 2, name, waits)
        require(counts[0] == 0 and counts[1] in strip_counts, name, counts)
        n_expect = counts[1]
    w0 = min(waits)
    # the macroblock loop: the nearest label above the waits that a later instruction branches back to
    labi, back = None, []
    for i in range(w0, -1, -1):
        m = re.match(r'^(\.LBB\d+_\d+):', L[i])
        if not m:
            continue
        back = [k for k, l in enumerate(L) if k > max(waits) and re.search(r's_c?branch\S*\s+' + re.escape(m.group(1)) + r'\b', l)]
        if back:
            labi = i
            break
    require(labi is not None, name)
    # (a) while the loads are in flight -- from the load block of the loop to the back edge, and from the loop head to
    #     the waits -- the compiler must not use v100-v123 (it may use them as scratch registers between a wait and
    #     the next load block: the record has been moved out, the loads are not issued yet)
    loop_loads = [i for i in asm_loads if i > max(waits)]
    require(len(loop_loads) == n_block, name, loop_loads)
    bad = [(i, l) for i, l in touched if loop_loads[0] <= i <= max(back) or labi <= i <= max(waits)]
    require(not bad, name, 'compiler code touches the prefetch registers while loads are in flight', bad[:5])
    pro = [i for i in asm_loads if i < labi]
    bad = [(i, l) for i, l in touched if pro and pro[0] <= i < labi]
    require(not bad, name, 'compiler code touches the prefetch registers behind the prologue loads', bad[:5])
    if wide:   # no vector load the compiler knows of inside the macroblock loop: it would wait for vmcnt(0) all over the step
        vis = [(i, L[i]) for i in range(labi, max(back) + 1)
               if re.match(r'\s+(global|flat|buffer)_load', L[i].split(';')[0]) and not _in_asm(L, i)]
        require(not vis, name, 'compiler-visible vector loads in the loop of a wide kernel', vis[:5])
    spills = [i for i in range(labi, max(back) + 1) if re.match(r'\s+scratch_', L[i])]
    require(not spills, name, 'scratch access inside the macroblock loop', spills[:3])
    in_loop = [i for i in asm_stores_at if labi <= i <= max(back)]
    if not os.environ.get("MVHP_CHECK_STRIP_COUNTS"):   # (measurement builds keep unreachable stores behind a false condition)
        require(len(in_loop) == len(asm_stores_at) == n_expect, name, len(in_loop), len(asm_stores_at), n_expect)
    return n_expect, len(asm_loads)


def main(path):
    text = open(path).read().split('\n')
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN4mvhp\d+recon_(quad|oct)_kernel\S+:', l)]
    require(starts)
    for s in starts:
        e = next(i for i in range(s, len(text)) if 's_endpgm' in text[i])
        name = text[s].split(':')[0]
        n, nd = check(text[s:e], name)
        print("ok", name, "stores per full strip", n, "asm loads", nd)


if __name__ == "__main__":
    main(sys.argv[1])
