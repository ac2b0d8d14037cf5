#!/usr/bin/env python3
"""Developer tool: workgroup barriers of one kernel that a wave can SKIP -- an `s_barrier` directly behind an
`s_cbranch_execz` / `s_cbranch_execnz` (a wave whose exec mask is empty at that point jumps over a barrier its sibling
waves execute: on this hardware that is a hang, not an error).
   hipcc -O3 --offload-arch=gfx950 ... --cuda-device-only -S -o k.s nl-partsol_amd/csrc/nlps_gpu.hip
   python tools/isa_barriers.py k.s <kernel-name-substring>"""
import re
import sys

txt = open(sys.argv[1]).read()
found = 0
for m in re.finditer(r"\.amdhsa_kernel (\S*%s\S*)" % re.escape(sys.argv[2]), txt):
    name = m.group(1)
    body = [l.strip() for l in txt[txt.find("\n" + name + ":"):m.start()].split("\n")]
    ins = [(i, l) for i, l in enumerate(body) if l and not l.startswith((";", ".")) or l.startswith(".LBB")]
    nb = sum(1 for _, l in ins if l.startswith("s_barrier"))
    bad = []
    for k, (i, l) in enumerate(ins):
        if not l.startswith("s_barrier"):
            continue
        prev = [x for x in ins[max(0, k - 3):k] if not x[1].startswith(".LBB")]
        if prev and prev[-1][1].startswith(("s_cbranch_execz", "s_cbranch_execnz")):
            bad.append((i, [p[1] for p in prev[-2:]]))
    print("%s: %d instructions, %d barriers, %d of them behind an exec-mask branch" % (name[:60], len(ins), nb, len(bad)))
    for i, p in bad:
        print("   line %d: %s ; s_barrier" % (i, " ; ".join(p)))
    found += 1
if not found:
    print("no kernel matches", sys.argv[2])
