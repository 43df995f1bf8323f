"""Summarise a tools/prof.sh output directory: per-kernel time stats + per-kernel PMC means."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    name = name.split("(")[0]
    return name[:70]


# ---- kernel stats
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    print("%-72s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:25]:
        print("%-72s %8s %12.1f %10.2f %7s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))

# ---- PMC: average counter value per dispatch, per kernel
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n== PMC (mean per dispatch)")
for k in sorted(acc, key=lambda k: -sum(acc[k].get("GRBM_GUI_ACTIVE", [0]))):
    if not any(s in k for s in ("msda", "linear", "chain", "gather", "triang", "pack", "add_ln", "mean_views", "class_head", "rowdot", "project")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-34s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
