#!/usr/bin/env python3
"""Developer tool: static instruction count per source line of one kernel, from an assembly listing made with
   hipcc -O3 --offload-arch=gfx950 ... -gline-tables-only --cuda-device-only -S -o build/kd/dev_g.s nl-partsol_amd/csrc/nlps_gpu.hip
   python tools/isa_lines.py build/kd/dev_g.s <mangled-kernel-name-prefix> [min-count]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
sym = sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
files = {}
for l in txt:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
start = next(i for i, l in enumerate(txt) if l.startswith(sym) and l.rstrip().endswith(':') or (l.startswith(sym) and ':' in l[:len(sym) + 200] and not l.startswith('\t')))
hist = collections.Counter()
cur = ("?", 0)
n = 0
for l in txt[start:]:
    s = l.strip()
    if s.startswith('.amdhsa_kernel') or s.startswith('.Lfunc_end'):
        break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
        continue
    hist[cur] += 1
    n += 1
print(n, "instructions")
byfile = collections.Counter()
for (f, l), c in hist.items():
    byfile[f] += c
print(dict(byfile))
for (f, l), c in sorted(hist.items()):
    if c >= minc:
        print("%-24s %5d %5d" % (f, l, c))
