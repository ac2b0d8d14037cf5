#!/usr/bin/env python3
"""Developer tool: instruction mix of one kernel by loop depth:  tools/isa_depth.py file.s kernel_name_substring [pattern ...]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
m = re.search(r"\.amdhsa_kernel (\S*%s\S*)" % re.escape(sys.argv[2]), txt)
name = m.group(1)
b = txt[txt.find("\n" + name + ":"):m.start()].split("\n")
pats = sys.argv[3:] or ["scratch_", "v_readlane", "v_writelane", "s_load", "global_load", "s_waitcnt", "ds_", "s_barrier"]
tot = Counter(); cnt = {p: Counter() for p in pats}
curdepth = 0; curhdr = None
for i, l in enumerate(b):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        j = i + 1; info = l
        while j < len(b) and b[j].strip().startswith(';'):
            info += b[j]; j += 1
        d = re.findall(r"Depth=(\d+)", info)
        curdepth = max(int(x) for x in d) if d else 0
        h = re.findall(r"Header=(BB\d+_\d+)", info)
        curhdr = h[-1] if h else (mm.group(1)[2:] if 'Loop Header' in info else None)
    if l.startswith("\t") and not l.strip().startswith((".", ";")):
        tot[(curdepth, curhdr)] += 1
        for p in pats:
            if p in l: cnt[p][(curdepth, curhdr)] += 1
print("%-22s %6s " % ("(depth, header)", "instr") + " ".join("%11s" % p[:11] for p in pats))
for k in sorted(tot, key=lambda x: (x[0], str(x[1]))):
    print("%-22s %6d " % (k, tot[k]) + " ".join("%11d" % cnt[p][k] for p in pats))
