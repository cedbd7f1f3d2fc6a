# round 5, GPU job 2: the whole GPU suite on the round's code (adaptive chunk windows, panel cap, input side at size, solo replay), the
# default bench line, adaptive against one-pass-per-chunk windows under paced links, a finer panel sweep for wide operands, the ALS step's kernels.
# usage: gpurun --timeout 1800 -- bash tools/gpu_jobs/r05_job2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job2
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/ -x -q -m gpu --durations=12 -s -k "input_side" > "$OUT/gputests_input_side.log" 2>&1; echo rc=$? >> "$OUT/gputests_input_side.log" )
tail -n 12 "$OUT/gputests_input_side.log"
( timeout 1000 python -m pytest tests/ -x -q -m gpu --durations=12 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 22 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
( timeout 500 python "$R/bench.py" --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
python - "$OUT/bench_n1.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"], o["phases_s"]))
for e in o["secondary"]:
    if "rank share, config 3" in e["workload"]:
        print(e["p"], e["chunks"], "held %.3f ms (%.1f%%, %d launches) | all landed %.3f ms (%.1f%%, %d) | solo wall %.3f ms (%.1f%%, %d)" % (
            e["held"]["kernel_ms"], 100 * e["held"]["frac"], e["held"]["launches"], e["held_all_landed"]["kernel_ms"], 100 * e["held_all_landed"]["frac"],
            e["held_all_landed"]["launches"], e["solo"]["wall_ms"], 100 * e["solo"]["frac_wall"], e["solo"]["launches"]))
    elif "error" in e:
        print("ERROR", e["workload"][:60], e["error"])
PY
for M in 1 0; do
  echo "== HNH_WINDOW_MERGE=$M" | tee -a "$OUT/overlap_adaptive_vs_static.log"
  HNH_WINDOW_MERGE=$M timeout 300 python "$R/tools/overlap_probe.py" --chunks "" --tapers "1,2,2,2,1,1;1,2,1" --pace 40,60,100,150,250 --iters 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/overlap_adaptive_vs_static.log"
done
( timeout 400 python "$R/tools/wide_panels.py" --r 384,512 --panels 3,4,5,6 --json "$OUT/wide_panels_fine.json" 2>&1 | grep -v amdgpu.ids > "$OUT/wide_panels_fine.log" )
( timeout 200 python "$R/tools/wide_panels.py" --r 256 --panels 3,4,5 2>&1 | grep -v amdgpu.ids >> "$OUT/wide_panels_fine.log" )
cat "$OUT/wide_panels_fine.log"
( timeout 300 python "$R/tools/als_profile.py" 2>&1 | grep -v amdgpu.ids > "$OUT/als_profile.log" ); cat "$OUT/als_profile.log"
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_als" -o stats -- python "$R/tools/als_profile.py" > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_als" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag r05_als_step --stats "$S" --out "$OUT" | tee "$OUT/als_kernel_stats_top.txt"
rm -rf "$R/gpurun_out/prof_als"
ls -la "$OUT"
