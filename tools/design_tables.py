#!/usr/bin/env python3
"""Generates the per-round kernel table of DESIGN.md section 5 from profiles/: average duration under rocprofv3
(rNN_*kernel_stats.csv), HBM bytes per launch from the PMC passes (rNN_hbm_traffic.md / rNN_step_counters.md), vector
instructions per 64 particles (rNN_sq_counters.md), and the fractions of the 8 TB/s HBM peak they give with the
algorithmic bytes of SURVEY 8d.  usage: python tools/design_tables.py  (prints markdown)"""
import csv
import glob
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
ALG = {"K2": 223, "K3": 586, "K5": 408}
PAT = {"K2": ("void k2_tile<3, true, 256", "void k2_tile<3, true>"), "K3": ("void k3_tile<3, 0, 1", "void k3_tile_lazy<3, 0"),
       "K5": ("void k5_tile<3", "void k5_tile_lazy<3")}


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for k, pats in PAT.items():
            if r["Name"].startswith(pats):
                tot, calls = out.get(k, (0.0, 0))
                out[k] = (tot + float(r["TotalDurationNs"]), calls + int(r["Calls"]))
    return {k: v[0] / v[1] / 1e3 for k, v in out.items() if v[1]}


def traffic(tag):
    out = {}
    for fn in ("%s_hbm_traffic.md" % tag, "%s_step_counters.md" % tag):
        p = os.path.join(ROOT, fn)
        if not os.path.exists(p):
            continue
        for l in open(p):
            c = [x.strip().strip("`") for x in l.split("|")]
            if len(c) < 5:
                continue
            for k, name in (("K2", "k2_tile"), ("K3", "k3_tile"), ("K5", "k5_tile")):
                if c[1].startswith(name) and "false, 256" not in c[1].replace("<3, false", "<3,false"):
                    try:
                        out[k] = float(c[4]) if fn.endswith("hbm_traffic.md") else float(c[6])
                    except ValueError:
                        pass
    return out


def valu(tag):
    p = os.path.join(ROOT, "%s_sq_counters.md" % tag)
    out = {}
    if os.path.exists(p):
        for l in open(p):
            c = [x.strip() for x in l.split("|")]
            if len(c) > 4 and c[1] in ("k2_tile", "k3_tile", "k5_tile"):
                out["K" + c[1][1]] = (c[2], c[4], c[3])
    return out


print("| round | kernel | avg us (rocprofv3) | algorithmic B / particle | HBM frac (algorithmic) | counter MB / launch | HBM frac (counter) | VALU / SALU / LDS instr per 64 particles |")
print("|---|---|---|---|---|---|---|---|")
for tag in ("r01", "r02", "r03", "r04"):
    g = glob.glob(os.path.join(ROOT, "%s_bench_kernel_stats.csv" % tag)) + glob.glob(os.path.join(ROOT, "%s_step_kernel_stats.csv" % tag))
    if not g:
        continue
    s, t, v = stats(g[0]), traffic(tag), valu(tag)
    for k in ("K2", "K3", "K5"):
        if k not in s:
            continue
        us = s[k]
        fa = ALG[k] * 1e6 / (us * 1e-6) / 8e12
        tb = t.get(k)
        print("| %s | %s | %.1f | %d | %.3f | %s | %s | %s |" % (
            tag, k, us, ALG[k], fa, "%.0f" % tb if tb else "-", "%.3f" % (tb * 1e6 / (us * 1e-6) / 8e12) if tb else "-",
            " / ".join(v[k]) if k in v else "-"))
