#!/usr/bin/env python3
"""Summarises rocprofv3 outputs (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes + the
8-B-per-lane calibration copy) into profiles/: usage  tools/summarize_pmc.py gpurun_out/prof_rXX  rXX"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")


def counters(d):
    f = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, "%s_bench_kernel_stats.csv" % tag))
for law in ("hencky", "dp"):  # kernel stats of tools/kbench.py --law ... (tools/profile.sh step 4)
    g = glob.glob(os.path.join(src, "stats_" + law, "*", "*kernel_stats.csv"))
    if g:
        shutil.copy(g[0], os.path.join(out, "%s_kbench_%s_kernel_stats.csv" % (tag, law)))
cf, cw = counters("calib_fetch"), counters("calib_write")
known = (1 << 27) * 8 / 1024.0  # KiB read and written by tools/hbm_calib.hip per launch
kf = [v for k, v in cf.items() if "copy8" in k][0]
kw = [v for k, v in cw.items() if "copy8" in k][0]
fetch_corr, write_corr = known / kf, known / kw
f, w = counters("fetch"), counters("write")
names = {"k_search": "search+activate", "k2_tile": "lists+newton+p2g_mass_mom",
         "k3_tile": "g2p_grad+stress+p2g_force", "k5_tile": "g2p_update"}
traffic, lines = {}, []
for kern, nice in names.items():
    fk = sum(v for k, v in f.items() if kern in k)
    wk = sum(v for k, v in w.items() if kern in k)
    b = (fk * fetch_corr + wk * write_corr) * 1024.0
    traffic[nice] = b
    lines.append("| %s | %.1f | %.1f | %.1f |" % (kern, fk * fetch_corr * 1024 / 1e6, wk * write_corr * 1024 / 1e6, b / 1e6))
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
with open(os.path.join(out, "%s_hbm_traffic.md" % tag), "w") as fh:
    fh.write("# HBM traffic per launch (%s), bench.py at 1 M particles\n\n" % tag)
    fh.write("Separate rocprofv3 passes `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (with `--kernel-trace`).\n")
    fh.write("Calibration on this code's access width (8 B/lane coalesced doubles, tools/hbm_calib.hip, 1 GiB read + "
             "1 GiB written per launch): FETCH_SIZE reads %.4f of the true bytes (correction x%.3f), WRITE_SIZE %.4f "
             "(x%.3f) — the gfx950 halving of MI355X_MICROARCH.md §HBM also holds for 8-B lanes.\n\n"
             % (1 / fetch_corr, fetch_corr, 1 / write_corr, write_corr))
    fh.write("| kernel | read MB | written MB | total MB |\n|---|---|---|---|\n" + "\n".join(lines) + "\n")
print(open(os.path.join(out, "%s_hbm_traffic.md" % tag)).read())
