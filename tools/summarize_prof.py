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
    if not any(s in k for s in ("msda", "samp_chain", "linear", "chain", "wreg", "gather", "triang", "pack", "add_ln", "mean_views", "class_head", "rowdot", "project", "pyramid", "f32s", "wgrad", "bin_")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-34s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))

# ---- MFMA kernels: achieved matrix throughput (counted MFMA ops x 512 FLOP / kernel-trace duration) against the
# dense bf16 peak, LDS bank-conflict share
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = float(r["AverageNs"]) * 1e-9
print("\n== MFMA kernels (per launch)")
print("%-44s %9s %12s %10s %12s" % ("kernel", "avg_us", "MFMA TFLOP/s", "% 2.5 PF", "LDS confl %"))
for k in sorted(acc, key=lambda k: -dur.get(k, 0)):
    c = acc[k]
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" not in c or k not in dur:
        continue
    mean = lambda n: sum(c[n]) / len(c[n]) if n in c and c[n] else float("nan")
    mops = mean("SQ_INSTS_VALU_MFMA_MOPS_BF16")
    if not mops > 0:
        continue
    tf = mops * 512.0 / dur[k] / 1e12
    confl = 100.0 * mean("SQ_LDS_BANK_CONFLICT") / max(mean("SQ_LDS_IDX_ACTIVE"), 1.0)      # share of the LDS-active cycles
    print("%-44s %9.2f %12.1f %10.1f %12.1f" % (k[:44], dur[k] * 1e6, tf, 100.0 * tf / 2500.0, confl))

# ---- machine-readable HBM-side traffic of the dominant kernel (read by bench.py -> roofline.traffic)
import json
for k in acc:
    if (k.startswith("msda_gsamp_pipe_kernel") or k.startswith("msda_gsamp_kernel")) and "FETCH_SIZE" in acc[k]:
        fetch_kb = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
        write_kb = sum(acc[k].get("WRITE_SIZE", [0])) / max(len(acc[k].get("WRITE_SIZE", [0])), 1)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from bench import sampler_source_hash
        # what the figure was measured on: bench.py quotes it only for the same kernel sources and configuration
        rec = {"kernel": k, "config": os.environ.get("MVG_PROF_CONFIG", "cfg2"), "dtype": os.environ.get("MVG_PROF_DTYPE", "bf16"),
               "queries": int(os.environ.get("MVG_PROF_QUERIES", "1024")), "valid_fraction": None,
               "inside": os.environ.get("MVG_PROF_INSIDE", "grid"),
               "src_sha256": sampler_source_hash(),
               "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
               "fetch_correction": 2.0,
               "note": "gfx950 rocprofv3: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B "
                       "(MI355X_MICROARCH.md, HBM section); calibrated in the same run on kernels with a known read "
                       "volume: pack_level (fp32 read once) FETCH/actual = 0.49, value projection (103 MB bf16 read once) 0.56",
               "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
        with open(os.path.join(out, "pmc_msda.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print("\nmsda traffic/launch: %.1f MB (2 x FETCH_SIZE %.1f MB + WRITE_SIZE %.1f MB)"
              % (rec["traffic_bytes_per_launch"] / 1e6, fetch_kb * 1024 / 1e6, write_kb * 1024 / 1e6))
