import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if "art::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]            # steady state
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("kernels", len(rows), "makespan us", (t1 - t0) / 1e3, "sum of durations us", busy / 1e3, "avg concurrency", busy / (t1 - t0))
per = collections.defaultdict(list)
for r in rows: per[r["Kernel_Name"].split("(")[0][-30:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items(): print(k, "n", len(v), "avg us", sum(v) / len(v), "min", min(v), "max", max(v))
qs = collections.Counter(r.get("Queue_Id", "?") for r in rows); print("queues used", dict(qs))
