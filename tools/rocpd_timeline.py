"""Development tool: the kernel dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite database) as a timeline — start and
duration in ms relative to the first dispatch of the window, queue, kernel name — to see what actually ran side by side.
    python tools/rocpd_timeline.py <results.db> [--last-ms 80] [--min-us 20]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--last-ms", type=float, default=80.0, help="only the dispatches of the last so many milliseconds of the run")
ap.add_argument("--min-us", type=float, default=20.0, help="leave out dispatches shorter than this")
ap.add_argument("--busy", action="store_true", help="also read the memory copies (--memory-copy-trace) and split the window into: a kernel running, "
                "only copies / fills running, nothing running (host enqueue and event waits) — where a call's time outside its kernels goes")
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
copies = []
if a.busy:
    for n2 in names:
        cols2 = [r[1] for r in cur.execute("pragma table_info('%s')" % n2)]
        low2 = {c.lower(): c for c in cols2}
        if "start" in low2 and "end" in low2 and ("copy" in n2.lower() or "memory" in n2.lower()) and "alloc" not in n2.lower():
            try:
                nm = low2.get("name") or low2.get("kind") or low2.get("direction")
                got = list(cur.execute("select %s, %s, %s from %s" % (low2["start"], low2["end"], nm if nm else "''", n2)))
            except sqlite3.Error:
                continue
            if got:
                copies = got
                print("copies from table/view %s: %d" % (n2, len(got)))
                break
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - a.last_ms * 1e6 and (r[1] - r[0]) >= a.min_us * 1e3]
if a.busy:
    def union(iv):
        iv = sorted(iv)
        out = []
        for s_, e_ in iv:
            if out and s_ <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e_)
            else:
                out.append([s_, e_])
        return out

    def total(iv):
        return sum(e_ - s_ for s_, e_ in iv)

    def minus(a_, b_):  # a_ \ b_, both unions
        out = []
        for s_, e_ in a_:
            cur_ = s_
            for bs, be in b_:
                if be <= cur_ or bs >= e_:
                    continue
                if bs > cur_:
                    out.append([cur_, bs])
                cur_ = max(cur_, be)
            if cur_ < e_:
                out.append([cur_, e_])
        return out
    w0 = t_end - a.last_ms * 1e6
    allk = [(max(r[0], w0), r[1]) for r in cur.execute("select %s, %s from %s" % (low["start"], low["end"], n)) if r[1] > w0]
    fills = [(s_, e_) for s_, e_, nm in [(max(r[0], w0), r[1], r[3]) for r in cur.execute("select %s, %s, %s, %s from %s" % (low["start"], low["end"], queue_col, name_col, n)) if r[1] > w0]
             if "fillBuffer" in str(nm) or "copyBuffer" in str(nm)]
    kern = union([x for x in allk if x not in set(fills)])
    cp = union([(max(c_[0], w0), c_[1]) for c_ in copies if c_[1] > w0] + fills)
    span = [[min([k[0] for k in kern] + [c_[0] for c_ in cp]), t_end]]
    only_copy = minus(cp, kern)
    idle = minus(minus(span, kern), cp)
    tot = total(span)
    print("window %.3f ms: a kernel running %.3f ms (%.1f %%), only copies / fills running %.3f ms (%.1f %%), nothing running %.3f ms (%.1f %%)" % (
        tot / 1e6, total(kern) / 1e6, 100.0 * total(kern) / tot, total(only_copy) / 1e6, 100.0 * total(only_copy) / tot, total(idle) / 1e6, 100.0 * total(idle) / tot))
t0 = rows[0][0]
queues = {q: i for i, q in enumerate(sorted({r[2] for r in rows}))}
print("table %s; %d dispatches; queues: %s" % (n, len(rows), queues))
for s, e, q, nm in rows:
    print("%9.3f ms  +%8.3f ms  q%-2d %s%s" % ((s - t0) / 1e6, (e - s) / 1e6, queues[q], "    " * queues[q], str(nm)[:70]))
