#!/usr/bin/env python3
"""Summarises the SQ counter passes of a bench.py run (rocprofv3 --kernel-trace --pmc <three counters>, one directory
per pass under gpurun_out/prof_sq) into profiles/<tag>_sq_counters.md:  tools/summarize_sq.py gpurun_out/prof_sq r01"""
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_search", "k2_tile", "k3_tile", "k5_tile"]
# the 3-D instantiations of the explicit step (Neo-Hookean bench cloud); *_lazy: the folded step
PREFIX = {"k_search": ("void k_search<3",), "k2_tile": ("void k2_tile<3, true",),
          "k3_tile": ("void k3_tile<3, 0, 1", "void k3_tile_lazy<3, 0"), "k5_tile": ("void k5_tile<3", "void k5_tile_lazy<3")}
WAVES = 1000000 / 64.0  # particle-waves of the bench workload
val, dur = {}, {}
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if r["Kernel_Name"].startswith(PREFIX[k]):
                # the last launch of every kernel wins (rows are in dispatch order)
                val[(k, r["Counter_Name"])] = float(r["Counter_Value"])
                dur[(k, r["Counter_Name"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
import json  # noqa: E402

rows, js = [], {}
for k in KERNELS:
    g = lambda c: val.get((k, c), float("nan"))  # noqa: E731
    d_ns = dur[(k, "SQ_INSTS_VALU")]
    busy = g("SQ_ACTIVE_INST_VALU") / (1024 * dur[(k, "SQ_ACTIVE_INST_VALU")] * 2.4 / 4.0)
    conf = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0.0
    issue = g("SQ_INSTS_VALU") * 4.0 / (1024 * 2.4 * d_ns)                       # fp64_issue_frac of bench.py
    lds = g("SQ_LDS_IDX_ACTIVE") / (256 * 2.4 * dur[(k, "SQ_LDS_IDX_ACTIVE")])  # lds_busy_frac of bench.py
    js[k] = {"valu_insts": g("SQ_INSTS_VALU"), "lds_idx_active": g("SQ_LDS_IDX_ACTIVE"), "lds_insts": g("SQ_INSTS_LDS"),
             "duration_us_under_pmc": d_ns / 1e3}
    rows.append("| %s | %.0f | %.0f | %.0f | %.0f %% | %.0f %% | %.2f | %.2f | %.0f |" % (
        k, g("SQ_INSTS_VALU") / WAVES, g("SQ_INSTS_LDS") / WAVES, g("SQ_INSTS_SALU") / WAVES, 100 * busy, 100 * conf,
        issue, lds, d_ns / 1e3))
json.dump(js, open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w"), indent=1)
text = """# SQ counters per launch (%s), bench.py at 1 M particles, Neo-Hookean

Separate `rocprofv3 --kernel-trace --pmc ...` passes (three counters each), last launch of every kernel.  `VALU busy` =
SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x kernel duration x 2.4 GHz / 4); instructions per particle-wave =
counter / 15 625 waves of 64 particles (one wave instruction serves 64 particles).  `FP64 issue frac` = SQ_INSTS_VALU x 4
cycles / (1024 SIMDs x 2.4 GHz x duration); `LDS busy frac` = SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the CUs) /
(256 CUs x 2.4 GHz x duration), both at the nominal 2.4 GHz (the chip ran these kernels at about 2.2 GHz, SQ_BUSY_CYCLES / 32
shader engines / duration).  The same numbers go to profiles/sq_counters.json, which bench.py reads.  Made by tools/summarize_sq.py.

| kernel | VALU instr / particle-wave | LDS instr | SALU instr | VALU busy | LDS bank-conflict / active cycles | FP64 issue frac | LDS busy frac | duration us (under PMC) |
|---|---|---|---|---|---|---|---|---|
%s
""" % (tag, "\n".join(rows))
open(os.path.join(ROOT, "profiles", "%s_sq_counters.md" % tag), "w").write(text)
print(text)
