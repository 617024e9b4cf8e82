#!/usr/bin/env python3
"""per-kernel statistics of a rocprofv3 run kept as a rocpd database (the default output format of rocprofv3 in ROCm 7): tools/rocpd_stats.py <results.db> [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
import subprocess
rows = list(cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"))
for name, n, avg, mn, mx, tot in rows:
    if flt and flt not in name: continue
    try: name = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: pass
    print(f"{name[:80]:80s} n={n:6d} avg_us={avg:9.2f} min={mn:8.2f} max={mx:8.2f} total_ms={tot:9.2f}")
