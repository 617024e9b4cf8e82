#!/usr/bin/env python3
"""start / end of the last N k_frame dispatches of a rocprofv3 --kernel-trace run of `bench.py --steps N --plain` (csv output):
   tools/burst_timeline.py <dir with *_kernel_trace.csv> N   -- times in us relative to the first of them"""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_frame" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
ends = sorted(int(r["End_Timestamp"]) for r in rows)
print("i  start_us  end_us  span_us  queue")
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{i:2d} {s / 1e3:9.1f} {e / 1e3:8.1f} {(e - s) / 1e3:8.1f}  {r.get('Queue_Id', '')}")
print("burst: first start -> last end %.1f us; completions at" % ((ends[-1] - t0) / 1e3), " ".join("%.0f" % ((e - t0) / 1e3) for e in ends))
