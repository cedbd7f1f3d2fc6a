# round 5, GPU job 7: the GPU suite at the last code (new: hnh_ipc_pull alignment classes), the bench line without the CPU baseline
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job7
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=6 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 14 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
( timeout 500 python "$R/bench.py" --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
python - "$OUT/bench_n1.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"], o["phases_s"]))
for e in o["secondary"]:
    if "rank share, config 3" in e["workload"]:
        print(e["p"], e["chunks"], "held k %.3f w %.3f | landed k %.3f w %.3f | solo k %.3f w %.3f (%d launches)" % (e["held"]["kernel_ms"], e["held"]["wall_ms"],
              e["held_all_landed"]["kernel_ms"], e["held_all_landed"]["wall_ms"], e["solo"]["kernel_ms"], e["solo"]["wall_ms"], e["solo"]["launches"]))
    if "error" in e:
        print("ERROR", e["workload"][:60], e["error"])
PY
