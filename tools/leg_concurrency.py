#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: the frame kernel's mean duration, frames per unit time and mean number in flight, for the static leg and for the
moving-model leg (the frames between the first and the last refit kernel), and what the refit kernels add:
   tools/leg_concurrency.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(t)))
refit = [(s, e, n) for s, e, n in rows if "k_refit_sub" in n or "k_wide_requant" in n]
fr = [(s, e) for s, e, n in rows if "k_frame<" in n]
first, last = (min(s for s, e, n in refit), max(e for s, e, n in refit)) if refit else (1 << 62, 1 << 62)


def stat(v, tag):
    if not v:
        return
    d = [(e - s) / 1e3 for s, e in v]
    span = (max(e for s, e in v) - min(s for s, e in v)) / 1e3
    print(f"{tag}: {len(v)} frames, mean duration {sum(d) / len(d):.1f} us, {span / len(v):.1f} us a frame, {sum(d) / span:.2f} in flight")


st = [(s, e) for s, e in fr if e < first]
stat(st[len(st) // 2:len(st) // 2 + 1000], "static (1 000 frames from the middle)")
mv = [(s, e) for s, e in fr if first <= s <= last]
stat(mv[len(mv) // 4:], "moving model (the last three quarters)")
if refit:
    for key in ("k_refit_sub", "k_wide_requant"):
        d = [(e - s) / 1e3 for s, e, n in refit if key in n]
        print(f"{key}: {len(d)} launches, mean {sum(d) / len(d):.1f} us, min {min(d):.1f}")
