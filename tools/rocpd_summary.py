#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel count / total / average / max, plus the wavefront
rounds of the last render (average logic / traversal kernel time, first and last rounds).   rocpd_summary.py results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n.split("(")[0][:70]].append((e - s) / 1e3)
print("%-72s %7s %12s %10s %10s" % ("kernel", "calls", "total ms", "avg us", "max us"))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-72s %7d %12.3f %10.1f %10.1f" % (n, len(v), sum(v) / 1e3, sum(v) / len(v), max(v)))
inits = [i for i, (n, s, e) in enumerate(rows) if "wf_init" in n]
if inits:
    seg = rows[inits[-1]:]
    lg = [(e - s) / 1e3 for n, s, e in seg if "wf_logic" in n]
    tr = [(e - s) / 1e3 for n, s, e in seg if "wf_trav" in n]
    print("last wavefront render: %d rounds, span %.2f ms, kernel time %.2f ms" % (len(lg), (seg[-1][2] - seg[0][1]) / 1e6, sum(e - s for n, s, e in seg) / 1e6))
    print("logic avg %.1f us, traversal avg %.1f us" % (sum(lg) / len(lg), sum(tr) / len(tr)))
    print("logic  first 16:", [round(x) for x in lg[:16]], "last 12:", [round(x) for x in lg[-12:]])
    print("trav   first 16:", [round(x) for x in tr[:16]], "last 12:", [round(x) for x in tr[-12:]])
