# round 5, GPU job 15: the bench line with the 2.5D sparse-replicate and 15d_fusion1 rank shares
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job15
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
( timeout 500 python "$R/bench.py" --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
python - "$OUT/bench_n1.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"], o["phases_s"]))
for e in o["secondary"]:
    if "solo" in e and "rank share, config 3" not in e["workload"]:
        print(e["workload"][:60], "| kernel %.3f ms wall %.3f ms frac_kernel %.3f frac_wall %.3f launches %d | %.1f s" % (e["solo"]["kernel_ms"], e["solo"]["wall_ms"],
              e["solo"].get("frac_kernel", 0), e["solo"]["frac_wall"], e["solo"]["launches"], e["seconds"]))
    if "error" in e:
        print("ERROR", e["workload"][:60], e["error"])
PY
