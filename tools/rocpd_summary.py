"""Summarise rocprofv3 rocpd sqlite output (kernel stats or PMC sums) as text for profiles/."""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
print(f"# source: {db}")
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
if rows:
    print("## kernel-trace --stats (durations in ns)")
    print(f"{'kernel':90s} {'calls':>6s} {'total_ns':>14s} {'avg_ns':>14s} {'pct':>7s}")
    for n, c, t, a, p in rows:
        print(f"{n[:90]:90s} {c:6d} {t*1e3:14.0f} {a*1e3:14.0f} {p:7.2f}")
try:
    rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
except sqlite3.OperationalError:
    rows = []
if rows:
    print("## PMC sums per kernel (sum over dispatches and over SE/XCD instances)")
    for k, c, v, n in rows:
        print(f"{k[:70]:70s} {c:28s} {v:22.0f}  dispatches={n}")
