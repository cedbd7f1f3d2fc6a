"""What every test that runs bench.py does with the ONE line it prints: the line is held to its bounds, the run's full record is what
the test then reads (benchlib/line.py: the line is the record's bounded form and names the record in `full_record`)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LINE_LIMIT = 8192  # bytes; the driver reads the result out of a bounded tail of the run's output (BENCH_r05: a 19 KB line, parsed: null)


def _no_constants(name):
    raise ValueError("not strict JSON: %s" % name)


def read_line(text, want_record=True):
    """The ONE line a run prints: at most LINE_LIMIT bytes, strict JSON (no NaN / Infinity), every string bounded, the contract's keys —
    checked for every line any test of this file sees.  Returns the run's FULL record, which the line names in `full_record`
    (benchlib/line.py: the line is the record's bounded form), so that the tests below keep reading every detail."""
    assert len(text.encode()) + 1 <= LINE_LIMIT, len(text)
    line = json.loads(text, parse_constant=_no_constants)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert "workload" in line["config"]

    def strings(v):
        if isinstance(v, str):
            yield v
        elif isinstance(v, dict):
            for k, x in v.items():
                yield k
                yield from strings(x)
        elif isinstance(v, list):
            for x in v:
                yield from strings(x)
    assert max(len(x) for x in strings(line)) <= 300
    if line.get("value") is not None:
        assert "roofline" in line and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    if not want_record:
        return line
    assert "full_record" in line, sorted(line)
    with open(line["full_record"] if os.path.isabs(line["full_record"]) else os.path.join(ROOT, line["full_record"])) as f:
        rec = json.load(f)
    for key in ("value", "n_gpus", "steps", "ms_per_step"):  # the line IS the record, shortened
        assert rec.get(key) == line.get(key), key
    return rec
