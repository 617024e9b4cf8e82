#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py's moving-model leg: the chain refit -> frame, launch by launch (the i-th batch kernel, the i-th crown, the i-th records
kernel and the i-th frame behind the first refit belong together): how long the boxes take, how long the frame then waits, how far apart the refits start.
   tools/refit_chain.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 0)) for r in csv.DictReader(open(t)))
sub = [r for r in rows if "k_refit_sub" in r[2]]
crown = [r for r in sub if r[3] == r[4]]            # one workgroup
batch = [r for r in sub if r[3] != r[4]]
rec = [r for r in rows if "k_wide_requant" in r[2]][1:]   # (the first is the cost of the tree as built)
first = batch[0][0]
frames = [r for r in rows if "k_frame<" in r[2] and r[0] >= first]
n = min(len(batch), len(crown), len(rec), len(frames))
lo = n // 4
B, C, R, F = (np.array([(r[0], r[1]) for r in v[lo:n]], dtype=np.float64) / 1e3 for v in (batch, crown, rec, frames))
print(f"{n - lo} refits and their frames (the last three quarters of the leg), microseconds:")
print(f"  batches {np.mean(B[:, 1] - B[:, 0]):.1f}   gap to the crown {np.mean(C[:, 0] - B[:, 1]):.1f}   crown {np.mean(C[:, 1] - C[:, 0]):.1f}   boxes start to end {np.mean(C[:, 1] - B[:, 0]):.1f}")
print(f"  frame starts {np.mean(F[:, 0] - C[:, 1]):.1f} after its boxes (median {np.median(F[:, 0] - C[:, 1]):.1f}), lasts {np.mean(F[:, 1] - F[:, 0]):.1f}")
print(f"  records start {np.mean(R[:, 0] - C[:, 1]):.1f} after the boxes, last {np.mean(R[:, 1] - R[:, 0]):.1f}")
print(f"  refits start {np.mean(np.diff(B[:, 0])):.1f} apart, frames end {np.mean(np.diff(F[:, 1])):.1f} apart")
print(f"  a refit starts {np.mean(B[8:, 0] - F[:-8, 1]):.1f} after the frame eight before it ended (its ring slot's previous frame)")
for back, name, arr, col in ((16, "frame 16 before ended", F, 1), (12, "frame 12 before ended", F, 1), (8, "frame 8 before ended", F, 1), (4, "crown 4 before ended", C, 1), (16, "records 16 before ended", R, 1), (1, "refit before started", B, 0)):
    d = B[back:, 0] - arr[:-back, col]
    print(f"  a refit starts {np.mean(d):8.1f} (p10 {np.percentile(d, 10):8.1f}, p90 {np.percentile(d, 90):8.1f}) after the {name}")
