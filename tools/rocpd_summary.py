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
    # dispatch durations of the same (counter-collecting) run, so that per-time ratios (clock = GRBM_GUI_ACTIVE / 8 / ns) can be formed
    print("## dispatch durations in the PMC run (ns; serialised dispatches)")
    q = ("select kernel_name, count(*), sum(duration), min(duration), max(duration) from "
         "(select distinct dispatch_id, kernel_name, duration from counters_collection) group by kernel_name")
    for k, n, s, lo, hi in cur.execute(q):
        print(f"{k[:70]:70s} dispatches={n:5d} total_ns={s:14.0f} min_ns={lo:12.0f} max_ns={hi:12.0f}")
