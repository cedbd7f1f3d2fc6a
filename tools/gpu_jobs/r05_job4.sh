# round 5, GPU job 4: where a SMALL call's time goes (config 1 as typed), API + kernel tables; the bench line with the larger config-4 rank share.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
python "$R/tools/small_call_probe.py" 2>&1 | grep -v amdgpu.ids | tee "$OUT/small_call.log"
python "$R/tools/small_call_probe.py" --alg 15d_fusion2 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/small_call.log"
python "$R/tools/small_call_probe.py" --p 1 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/small_call.log"
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d "$R/gpurun_out/prof_small" -o small -- python "$R/tools/small_call_probe.py" --iters 500 > "$OUT/small_call_traced.log" 2>&1
tail -2 "$OUT/small_call_traced.log"
find "$R/gpurun_out/prof_small" -name "*stats*.csv" | head
for f in $(find "$R/gpurun_out/prof_small" -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -25 "$f"; cp "$f" "$OUT/"; done
S=$(find "$R/gpurun_out/prof_small" -name "*_results.db" | head -1)
if [ -n "$S" ]; then python - "$S" <<'PY' | tee "$OUT/small_call_api_table.txt"
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([n for n in names if "top" in n or "summary" in n or "api" in n.lower()][:40])
for view in ("top_kernels", "top_hip_api", "hip_api_summary", "top"):
    if view in names:
        print("==", view)
        for row in list(cur.execute("select * from %s" % view))[:30]:
            print(row)
PY
fi
rm -rf "$R/gpurun_out/prof_small"
( timeout 500 python "$R/bench.py" --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
python - "$OUT/bench_n1.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"], o["phases_s"]))
for e in o["secondary"]:
    if "rank share, config" in e["workload"] or "config 1" in e["workload"]:
        print(json.dumps({k: v for k, v in e.items() if k != "workload"})[:700])
    if "ALS" in e["workload"]:
        print({k: e.get(k) for k in ("ms", "frac_whole_step", "frac_with_cg_row_streams")})
    if "error" in e:
        print("ERROR", e["workload"][:60], e["error"])
PY
ls -la "$OUT"
