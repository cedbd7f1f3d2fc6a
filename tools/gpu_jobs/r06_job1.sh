# round 6, GPU job 1: the driver's own bench command (is the line parseable: size, strict JSON), GPU tests with the observed ALS errors
# logged, kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job1
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.stderr" ) 2> "$OUT/bench_driver_cmd.time"
cp bench_secondary.json "$OUT/bench_driver_cmd_full_record.json" 2>/dev/null
python3 - "$OUT/bench_driver_cmd.json" <<'PY' | tee "$OUT/bench_line_check.txt"
import json, sys
raw = open(sys.argv[1]).read()
lines = [l for l in raw.splitlines() if l.strip()]
print("stdout lines:", len(lines), "bytes of the line:", len(lines[-1]) + 1)
def bad(x): raise ValueError(x)
d = json.loads(lines[-1], parse_constant=bad)
print({k: len(json.dumps(v)) for k, v in d.items()})
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("cpu_baseline", d.get("cpu_baseline"))
print("phases_s", d.get("phases_s"))
PY
wc -c "$OUT/bench_driver_cmd.stderr"
export HNH_OBSERVED_LOG="$OUT/observed_errors.jsonl"
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > "$OUT/gputests.log" 2>&1
tail -6 "$OUT/gputests.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o cfg2 -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-traffic > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.stderr"
DB=$(find "$OUT/prof" -name "*_results.db" | head -1)
python3 "$R/tools/rocpd_summary.py" --tag r06_cfg2 --stats "$DB" --out "$OUT" 2>&1 | tee "$OUT/cfg2_kernel_stats.txt"
rm -rf "$OUT/prof"
