#!/usr/bin/env python3
"""Developer tool: table of the SQ counters collected by tools/sq_profile.sh (last launch of every kernel)."""
import csv
import glob
import os
import sys

src = sys.argv[1]
val, dur = {}, {}
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        val[(k, r["Counter_Name"])] = float(r["Counter_Value"])
        dur[k] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
kernels = sorted({k for k, _ in val}, key=lambda k: -dur[k])[:8]
counters = sorted({c for _, c in val})
print("%-26s" % "counter" + "".join("%18s" % k[:17] for k in kernels))
print("%-26s" % "duration us" + "".join("%18.1f" % (dur[k] / 1e3) for k in kernels))
for c in counters:
    print("%-26s" % c + "".join("%18.4g" % val.get((k, c), float("nan")) for k in kernels))
