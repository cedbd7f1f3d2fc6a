"""The printed line and the full record.  The driver parses ONE JSON line out of a bounded tail of the run's output: a line that
outgrows that tail is a run without a result (round 5: a 19 KB line, `parsed: null`).  So what goes to stdout is the contract's keys
plus `roofline`, `cpu_baseline`, `check`, `phases_s` and a three-column table of the secondary workloads — every string cut to
STRING_LIMIT characters, the whole line at most LINE_LIMIT bytes, enforced here for every line any code path prints — and everything
else (the prose, the secondary entries in full, the search's table) goes to a side file next to it, named in the line.
The reference's own record is of this size (benchmark_dist.cpp:144-162: elapsed, throughput, alg_info, perf_stats)."""
import json
import os

LINE_LIMIT = 8192     # bytes, newline included: never exceeded (asserted)
LINE_TARGET = 6144    # optional parts are shed, in SHED_ORDER, until the line is below this
STRING_LIMIT = 80
LONG_STRINGS = {"workload": 200, "sample": 120, "error": 300, "incomplete": 200}  # keys whose text is the point
RECORD_NAME = "bench_secondary.json"

# what may go when a line is still too long (several GPUs: the search's tables), least needed first; dotted paths
SHED_ORDER = ("line_of_rank0", "config.route_tuning_failures", "config.budget_stops", "config.transport_trials", "preflight", "config.route_tuning_ms_per_step",
              "config.ranks", "secondary", "phases", "exit_codes", "phases_s", "check")


AMENDABLE = ("incomplete", "exit_codes", "phases", "error", "failed_rank", "phase")  # what a forwarding launcher adds to a worker's line

CORE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "cpu_baseline", "error", "incomplete", "backend", "full_record")


def record_path():
    """Where the full record goes: $HNH_BENCH_RECORD, else ./bench_secondary.json (the directory the run was started in)."""
    return os.environ.get("HNH_BENCH_RECORD") or os.path.join(os.getcwd(), RECORD_NAME)


def _sig(x, digits=6):
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")):
        return float("%.*g" % (digits, x))
    return x


def _cut(s, limit):
    return s if len(s) <= limit else s[:limit - 3] + "..."


def _shorten(v, key=None):
    """strings cut to their limit, non-finite floats to null (strict JSON), recursively"""
    if isinstance(v, str):
        return _cut(v, LONG_STRINGS.get(key, STRING_LIMIT))
    if isinstance(v, float):
        return v if (v == v and v not in (float("inf"), float("-inf"))) else None
    if isinstance(v, dict):
        return {_cut(str(k), STRING_LIMIT): _shorten(x, k) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_shorten(x, key) for x in v]
    return v


def secondary_row(e):
    """One secondary entry as {"id", "ms", "frac"} (+ "frac_wall" / "launches" for a rank's share, "ok" when a check failed)."""
    row = {"id": e.get("id") or _cut(e.get("workload", "?"), 24)}
    if "error" in e:
        row["error"] = _cut(str(e["error"]), 60)
        return row
    if "fused" in e:  # the width entries: the fused pass is the row, the un-fused pair beside it
        row.update(ms=e["fused"]["ms"], frac=e["fused"]["frac"])
        for k in ("sddmm", "spmm"):
            if k in e:
                row[k] = e[k]["frac"]
    elif "solo" in e:  # one rank's share: wall and kernels of the rank alone
        s = e["solo"]
        best = e.get("held_all_landed", s)
        row.update(ms=s["wall_ms"], frac=best.get("frac_kernel", best.get("frac")), frac_wall=s.get("frac_wall"), launches=s.get("launches"))
    else:
        row["ms"] = e.get("ms")
        row["frac"] = next((e[k] for k in ("frac", "frac_whole_step", "frac_all_ranks_on_this_one_gpu") if k in e), None)
    ok = (e.get("check") or {}).get("ok")
    if ok is False:
        row["ok"] = False
    return {k: _sig(v, 4) for k, v in row.items()}


def _pop_path(d, path):
    keys = path.split(".")
    for k in keys[:-1]:
        d = d.get(k)
        if not isinstance(d, dict):
            return False
    return d.pop(keys[-1], None) is not None


def compact(full, record=None):
    """The line that is printed for the full record `full` (a dict): same keys, bounded size."""
    out = dict(full)
    if isinstance(out.get("secondary"), list) and any("workload" in e for e in out["secondary"]):  # (full entries, not yet the table)
        sec = out["secondary"]
        out["secondary"] = [secondary_row(e) for e in sec]
        bad = [r["id"] for r in out["secondary"] if r.get("ok") is False or "error" in r]
        out["secondary_checks"] = {"entries": len(sec), "failed": bad}
    tuning = (out.get("config") or {}).get("route_tuning_ms_per_step")
    if isinstance(tuning, dict) and len(tuning) > 8:  # the search's table: the five fastest candidates, the rest in the record
        out["config"] = dict(out["config"])
        ranked = sorted(((v, k) for k, v in tuning.items() if v is not None))
        out["config"]["route_tuning_ms_per_step"] = dict({k: v for v, k in ranked[:5]}, candidates=len(tuning), failed=sum(1 for v in tuning.values() if v is None))
    out = _shorten(out)
    if record:
        out["full_record"] = record
    shed = []
    for path in SHED_ORDER:
        if len(json.dumps(out)) + 1 <= LINE_TARGET:
            break
        if _pop_path(out, path):
            shed.append(path)
            out["shed"] = shed  # (what was left out of the line for its size; it is in the record)
    return out


def render(obj):
    """(line, record path or None): the bytes for stdout, newline included, and where the full record went."""
    if isinstance(obj, str):
        line = obj
        path = None
    else:
        path = None
        try:
            p = record_path()
            rec = obj
            if "full_record" in obj:
                # a line that went through here before (bench.py's launcher forwarding rank 0's line with its own remarks): the record
                # of the run is the worker's — amended, not replaced by the shortened form
                obj = {k: v for k, v in obj.items() if k not in ("full_record", "shed")}
                try:
                    with open(p) as f:
                        rec = json.load(f)
                    rec.update({k: obj[k] for k in AMENDABLE if k in obj})
                except (OSError, ValueError):
                    rec = obj
            tmp = "%s.%d" % (p, os.getpid())
            with open(tmp, "w") as f:
                json.dump(rec, f, indent=1, default=str)
                f.write("\n")
            os.replace(tmp, p)
            path = p
        except OSError:
            pass  # (a read-only working directory costs the record, never the line)
        shown = path if (path is None or os.environ.get("HNH_BENCH_RECORD")) else RECORD_NAME  # (relative to where the run was started)
        line = json.dumps(compact(obj, shown), allow_nan=False)
    data = (line + "\n").encode()
    if len(data) > LINE_LIMIT and not isinstance(obj, str):
        # (not reachable with today's keys: SHED_ORDER ends far below the target) — the contract's keys and the numbers of the two objects
        core = {k: v for k, v in compact(obj, shown).items() if k in CORE_KEYS}
        for k in ("roofline", "cpu_baseline"):
            if isinstance(core.get(k), dict):
                core[k] = {a: b for a, b in core[k].items() if not isinstance(b, (str, dict, list)) or a in ("bound", "unit", "kind")}
        core["config"] = {"workload": (obj.get("config") or {}).get("workload", "")[:200]}
        core["shed"] = ["everything but the contract's keys"]
        data = (json.dumps(core, allow_nan=False) + "\n").encode()
    assert len(data) <= LINE_LIMIT, "bench.py: the result line is %d bytes, more than the %d the driver's tail is known to hold" % (len(data), LINE_LIMIT)
    return data, path
