"""Development tool: the kernel dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite database) as a timeline — start and
duration in ms relative to the first dispatch of the window, queue, kernel name — to see what actually ran side by side.
    python tools/rocpd_timeline.py <results.db> [--last-ms 80] [--min-us 20]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--last-ms", type=float, default=80.0, help="only the dispatches of the last so many milliseconds of the run")
ap.add_argument("--min-us", type=float, default=20.0, help="leave out dispatches shorter than this")
a = ap.parse_args()
con = sqlite3.connect(a.db)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
cand = None
for n in names:
    cols = [r[1] for r in cur.execute("pragma table_info('%s')" % n)]
    low = [c.lower() for c in cols]
    if "start" in low and "end" in low and any("queue" in c for c in low) and any(c in ("name", "kernel_name") or "kernel" in c for c in low):
        cand = (n, cols)
        if n.lower() in ("kernels", "kernel_dispatch"):
            break
if cand is None:
    raise SystemExit("no dispatch table found among: %s" % ", ".join(names))
n, cols = cand
low = {c.lower(): c for c in cols}
name_col = low.get("name") or low.get("kernel_name") or next(c for c in cols if "kernel" in c.lower() and "id" not in c.lower())
queue_col = next(c for c in cols if "queue" in c.lower())
rows = list(cur.execute("select %s, %s, %s, %s from %s order by %s" % (low["start"], low["end"], queue_col, name_col, n, low["start"])))
if not rows:
    raise SystemExit("no dispatches")
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - a.last_ms * 1e6 and (r[1] - r[0]) >= a.min_us * 1e3]
t0 = rows[0][0]
queues = {q: i for i, q in enumerate(sorted({r[2] for r in rows}))}
print("table %s; %d dispatches; queues: %s" % (n, len(rows), queues))
for s, e, q, nm in rows:
    print("%9.3f ms  +%8.3f ms  q%-2d %s%s" % ((s - t0) / 1e6, (e - s) / 1e6, queues[q], "    " * queues[q], str(nm)[:70]))
