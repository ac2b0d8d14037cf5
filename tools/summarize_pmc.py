#!/usr/bin/env python3
"""Summarises the rocprofv3 outputs of tools/profile.sh into profiles/:  tools/summarize_pmc.py gpurun_out/prof_rXX rXX [parts]
Per part (step | residual | twod | dp): the kernel-stats csv is copied as profiles/rXX_<part>_kernel_stats.csv, the
FETCH_SIZE / WRITE_SIZE passes (corrected by the calibration copy of the same box, tools/hbm_calib.hip) and the SQ counter
groups become profiles/rXX_<part>_counters.md -- one row per kernel that takes more than 1 % of the part's kernel time:
HBM bytes read / written per launch, VALU / SALU / LDS instructions per launch, LDS bank-conflict share, duration under
the counters.  JSON for bench.py: hbm_traffic.json (step), hbm_traffic_2d.json (twod); sq_counters.json comes from
tools/summarize_sq.py gpurun_out/prof_rXX/step_sq rXX."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
parts = sys.argv[3:] or ["step"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
STATS_DIR = {"step": ["step_stats"], "residual": ["residual_stats"], "twod": ["twod_stats"],
             "dp": ["kbench_dp_stats", "kbench_hencky_stats"]}
PMC_NAME = {"step": "step", "residual": "residual", "twod": "twod", "dp": "kbench_dp"}
NICE = {"k_search": "search+activate", "k2_tile": "lists+newton+p2g_mass_mom", "k3_tile": "g2p_grad+stress+p2g_force",
        "k5_tile": "g2p_update"}


def short(name):
    """`void k3_tile_lazy<3, 0>(PView, ...)` -> `k3_tile_lazy<3, 0>`"""
    n = name.strip('"')
    n = re.sub(r"^void ", "", n)
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def counters(d):
    """-> {kernel: {counter: (mean value per launch, mean duration ns, launches)}}"""
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            res[short(r["Kernel_Name"])][r["Counter_Name"]].append(
                (float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    return {k: {c: (sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v)) for c, v in cs.items()}
            for k, cs in res.items()}


def calib():
    cf, cw = counters("calib_fetch"), counters("calib_write")
    known = (1 << 27) * 8 / 1024.0  # KiB read and written by tools/hbm_calib.hip per launch
    kf = [v["FETCH_SIZE"][0] for k, v in cf.items() if "copy8" in k][0]
    kw = [v["WRITE_SIZE"][0] for k, v in cw.items() if "copy8" in k][0]
    return known / kf, known / kw


fetch_corr, write_corr = calib()
for part in parts:
    # 1. kernel stats
    top, total = {}, 0.0
    for i, sd in enumerate(STATS_DIR[part]):
        g = glob.glob(os.path.join(src, sd, "**", "*kernel_stats.csv"), recursive=True)
        if not g:
            continue
        name = "%s_%s_kernel_stats.csv" % (tag, part if len(STATS_DIR[part]) == 1 else sd.replace("_stats", ""))
        shutil.copy(g[0], os.path.join(out, name))
        if i == 0:
            for r in csv.DictReader(open(g[0])):
                top[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"]))
                total += float(r["TotalDurationNs"])
    pm = PMC_NAME[part]
    f, w, sq = counters(pm + "_fetch"), counters(pm + "_write"), counters(pm + "_sq")
    rows, traffic = [], {}
    for k, (avg, calls, tot) in sorted(top.items(), key=lambda kv: -kv[1][2]):
        if tot < 0.01 * total:
            continue
        rd = f.get(k, {}).get("FETCH_SIZE", (float("nan"),))[0] * fetch_corr * 1024.0
        wr = w.get(k, {}).get("WRITE_SIZE", (float("nan"),))[0] * write_corr * 1024.0
        s = sq.get(k, {})
        g = lambda c: s.get(c, (float("nan"), float("nan"), 0))  # noqa: E731
        conf = g("SQ_LDS_BANK_CONFLICT")[0] / g("SQ_LDS_IDX_ACTIVE")[0] if g("SQ_LDS_IDX_ACTIVE")[0] else float("nan")
        rows.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.2f | %.2f | %.2f | %.0f %% | %.1f |" % (
            k, calls, avg / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, g("SQ_INSTS_VALU")[0] / 1e6, g("SQ_INSTS_SALU")[0] / 1e6,
            g("SQ_INSTS_LDS")[0] / 1e6, 100 * conf, g("SQ_INSTS_VALU")[1] / 1e3))
        for pat, nice in NICE.items():
            if k.startswith(pat) and part in ("step", "twod"):
                traffic[nice] = traffic.get(nice, 0.0) + rd + wr
    if part == "step":
        json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
    if part == "twod":
        json.dump(traffic, open(os.path.join(out, "hbm_traffic_2d.json"), "w"), indent=1)
    text = """# Counters per launch (%s, part `%s` of tools/profile.sh), 1 M particles

Separate rocprofv3 passes (`--kernel-trace --stats`; `--pmc FETCH_SIZE`; `--pmc WRITE_SIZE`; SQ groups of three).  HBM bytes
are corrected by the calibration copy of the same box (tools/hbm_calib.hip, 8-B lanes, 1 GiB in + 1 GiB out): FETCH_SIZE
x%.3f, WRITE_SIZE x%.3f.  avg us: `AverageNs` of the stats pass; `us (pmc)`: duration under the SQ counter pass.  Kernels
above 1 %% of the part's kernel time.  Instruction counts are wave instructions per launch (/ 15 625 = per 64 particles).

| kernel | calls | avg us | read MB | written MB | total MB | VALU M instr | SALU M | LDS M | LDS bank-conflict / active | us (pmc) |
|---|---|---|---|---|---|---|---|---|---|---|
%s
""" % (tag, part, fetch_corr, write_corr, "\n".join(rows))
    open(os.path.join(out, "%s_%s_counters.md" % (tag, part)), "w").write(text)
    print(text)
