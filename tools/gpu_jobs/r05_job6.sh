# round 5, last GPU job: the round's final code — smoke, the whole GPU suite, the default bench line WITH the CPU baseline, kernel stats of the same
# command (roofline.avg_launch_ms must agree), eight and two processes on this one GPU through bench.py's product path (frac_step vs frac_kernel, phases_s, budget).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job6
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee "$OUT/smoke.log"
( timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=10 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 18 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
( timeout 700 python "$R/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
python - "$OUT/bench_n1.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f traffic/alg %.3f  cpu %.3e (%s cores)  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"],
      (o["roofline"]["traffic"] or 0) / o["roofline"]["algorithmic_bytes_per_launch"], o["cpu_baseline"]["value"] or 0, o["cpu_baseline"]["cores"], o["phases_s"]))
print("avg_launch_ms", o["roofline"]["avg_launch_ms"])
bad = [e for e in o["secondary"] if "error" in e or ("check" in e and not e["check"].get("ok", True))]
print("secondary entries", len(o["secondary"]), "bad", [(e["workload"][:50], e.get("error")) for e in bad])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_stats" -o stats -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-check --no-secondary > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_stats" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag r05_final_cfg2 --stats "$S" --out "$OUT"
rm -rf "$R/gpurun_out/prof_stats"
( timeout 600 python "$R/bench.py" --gpus 8 --no-tune --steps 3 --warmup 1 > "$OUT/bench_8proc_one_gpu.json" 2> "$OUT/bench_8proc_one_gpu.stderr"; echo "bench8 rc=$?" )
python - "$OUT/bench_8proc_one_gpu.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
r = o.get("roofline", {})
print("N=8 on one GPU: value %s ms %s frac %s frac_kernel %s frac_step %s exposed_comm_ms %s cpu_baseline %s phases %s" % (o.get("value"), o.get("ms_per_step"), r.get("frac"),
      r.get("frac_kernel"), r.get("frac_step"), r.get("exposed_comm_ms"), (o.get("cpu_baseline") or {}).get("value"), o.get("phases_s")), o.get("error"))
PY
( timeout 600 python "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --logm 18 --budget-s 240 > "$OUT/bench_2proc_one_gpu_budget240.json" 2> "$OUT/bench_2proc_one_gpu.stderr"; echo "bench2 rc=$?" )
python - "$OUT/bench_2proc_one_gpu_budget240.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("N=2 on one GPU, budget 240: value %s frac %s phases %s stops %s tuned %d transport_trials %s" % (o.get("value"), o.get("roofline", {}).get("frac"), o.get("phases_s"),
      o.get("config", {}).get("budget_stops"), len(o.get("config", {}).get("route_tuning_ms_per_step") or {}), o.get("config", {}).get("transport_trials")), o.get("error"))
PY
ls -la "$OUT"
