#!/usr/bin/env python3
"""Static check of the quad kernel's ISA: between the prefetch loads (inline asm) and the guarded wait at the top
of the next macroblock step, no instruction may read or write the prefetch registers, and exactly VM_STORES
vector-memory instructions must be issued.  usage: check_prefetch_hazard.py file.s"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def check(L, name):
    """L = the lines of one kernel."""
    waits = [i for i, l in enumerate(L) if re.search(r's_waitcnt vmcnt\(\d+\)', l) and 'v_mov_b64 ' in L[i + 1]]
    assert len(waits) == 3, (name, waits)
    counts = sorted(int(re.search(r'vmcnt\((\d+)\)', L[w]).group(1)) for w in waits)
    assert counts[0] == 0 and counts[2] == counts[1] + 2, (name, counts)
    n_expect = counts[2]
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
    assert labi is not None, name
    loads = [i for i, l in enumerate(L) if 'global_load_dwordx2' in l and i > max(waits)]
    assert len(loads) == 12, (name, len(loads))
    dest = set()
    for i in loads:
        dest |= regs(L[i].split(',')[0])
    bad, asm_stores = [], 0
    in_asm = False
    for i in list(range(loads[-1] + 1, max(back) + 1)) + list(range(labi, w0)):
        raw = L[i]
        if '#ASMSTART' in raw:
            in_asm = True
        if '#ASMEND' in raw:
            in_asm = False
        l = raw.split(';')[0]
        if not l.strip() or l.strip().startswith('.'):
            continue
        if in_asm and re.match(r'\s+global_store', l):
            asm_stores += 1
        if regs(l) & dest:
            bad.append((i, l))
    for i in range(loads[0], loads[-1] + 1):
        l = L[i].split(';')[0]
        if 'global_load' in l or not l.strip():
            continue
        if regs(l) & dest:
            bad.append((i, l))
    # register spills (scratch_*) are vector-memory operations too: none may sit inside the macroblock loop
    spills = [i for i in range(labi, max(back) + 1) if re.match(r'\s+scratch_', L[i])]
    assert not spills, (name, 'scratch access inside the macroblock loop', spills[:3])
    # the two wait blocks themselves read the prefetch registers (that is their job); nothing else may
    bad = [(i, l) for i, l in bad if not any(w < i <= w + 12 for w in waits)]
    assert not bad, (name, bad[:5])
    # every asm store of the loop lies on the path of the step that parks nothing: pair flush + chroma flush
    assert asm_stores == n_expect, (name, asm_stores, n_expect)
    return n_expect, len(dest)


def main(path):
    text = open(path).read().split('\n')
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN4mvhp17recon_quad_kernel\S+:', l)]
    assert starts
    for s in starts:
        e = next(i for i in range(s, len(text)) if 's_endpgm' in text[i])
        name = text[s].split(':')[0]
        n, nd = check(text[s:e], name)
        print("ok", name, "stores", n, "prefetch registers", nd)


if __name__ == "__main__":
    main(sys.argv[1])
