"""Per-kernel timeline of the last forward in a rocprofv3 kernel trace (start/end relative to the step's first
kernel, stream/queue) -- shows what the side-stream kernels overlap with.  tools/timeline.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: starts at its pack launch (one mvg_pack_pyramid launch per forward; three pack_level launches in older builds)
idx = [i for i, r in enumerate(rows) if "pack_pyramid" in r["Kernel_Name"]]
if idx:
    start = idx[-1]
else:
    idx = [i for i, r in enumerate(rows) if "pack_level" in r["Kernel_Name"]]
    if len(idx) < 3:
        sys.exit("timeline.py: no pack_pyramid / pack_level launch in %s" % f)
    start = idx[-3]
t0 = int(rows[start]["Start_Timestamp"])
end = max(i for i, r in enumerate(rows) if "triangulate" in r["Kernel_Name"] or "finish_layer" in r["Kernel_Name"])
print("# one forward, us relative to the first pack kernel; q = HSA queue (main / side streams)")
print("%-42s %-4s %8s %8s  %6s" % ("kernel", "q", "start", "end", "dur"))
for r in rows[start:end + 1]:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:42]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-42s q%-3s %8.1f %8.1f  %6.1f" % (n, r.get("Queue_Id", "?"), s, e, e - s))
