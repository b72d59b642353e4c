#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 results.db: python tools/pmc_db.py DB [kernel-substring] [bytes-per-dispatch]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
per = float(sys.argv[3]) if len(sys.argv) > 3 else 0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tabs if 'pmc_event' in t][0]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ip = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = f"select s.kernel_name, p.name, sum(e.value), count(distinct k.id) from {pmc} e join {kd} k on e.event_id=k.event_id join {ip} p on e.pmc_id=p.id join {ks} s on k.kernel_id=s.id group by 1,2"
for name, ctr, val, n in cur.execute(q):
    if pat in name:
        extra = f"  {val / (per * n):10.3f} per byte" if per else ""
        print(f"{name.split('(')[0][:40]:40s} {ctr:26s} {val:14.6g}  dispatches {n}{extra}")
