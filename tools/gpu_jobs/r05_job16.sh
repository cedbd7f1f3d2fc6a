# round 5, GPU job 16: bench.py's multi-process paths on one GPU after the search gained its one-pass-per-chunk candidate
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job16
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 600 python -m pytest tests/test_multigpu_gpu.py -x -q -k "bench" --durations=5 > "$OUT/gputests_bench_multiproc.log" 2>&1; echo rc=$? >> "$OUT/gputests_bench_multiproc.log" )
tail -n 10 "$OUT/gputests_bench_multiproc.log"
cd /tmp
( timeout 300 python "$R/bench.py" --gpus 4 --steps 3 --warmup 1 --logm 18 > "$OUT/bench_4proc.json" 2>/dev/null; echo rc=$? )
python - "$OUT/bench_4proc.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
t = o["config"]["route_tuning_ms_per_step"]
for k in sorted(t, key=lambda k: (t[k] is None, t[k])):
    print("%8.3f  %s" % (t[k] or -1, k))
print(o["config"]["mesh_chunks"], o["phases_s"])
PY
