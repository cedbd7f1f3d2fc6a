#!/bin/bash
# Reproduces the headline evidence under profiles/ on a GPU box (run through gpurun from the repository root):
#   the bench line (which collects its own FETCH_SIZE / WRITE_SIZE passes live), rocprofv3 kernel-trace stats of the same
#   command, and — as an independent cross-check of the live figure — the two PMC passes run from here (each in its own run,
#   never combined with a trace domain).  Exports land in gpurun_out/profiles_export/ (copy them into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01_final}
OUT=$R/gpurun_out/profiles_export
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/${TAG}_bench_n1.stderr"
tail -c 600 "$OUT/${TAG}_bench_n1.json"
# the profiled runs use the CU setting the bench run measured to be faster (no second search: their kernel averages then belong
# to exactly the configuration that was timed)
CUS=$(python -c "import json,sys; print(json.load(open(sys.argv[1]))['config'].get('comm_cus', 0))" "$OUT/${TAG}_bench_n1.json" 2>/dev/null || echo 0)
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_stats" -o stats -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --no-check --comm-cus "$CUS" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$R/gpurun_out/prof_fetch" -o fetch -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-check --comm-cus "$CUS" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$R/gpurun_out/prof_write" -o write -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-check --comm-cus "$CUS" > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_stats" -name "*_results.db" | head -1)
F=$(find "$R/gpurun_out/prof_fetch" -name "*_results.db" | head -1)
W=$(find "$R/gpurun_out/prof_write" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag "${TAG}_cfg2" --stats "$S" --pmc "$F" "$W" --workload-key er20_ef96_r128_n1 --out "$OUT"
rm -rf "$R/gpurun_out/prof_stats" "$R/gpurun_out/prof_fetch" "$R/gpurun_out/prof_write"
ls -la "$OUT"
