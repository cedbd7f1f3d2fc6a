# round 5, GPU job 10: the driver's own N = 1 command at the round's last code
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job10
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
T0=$(date +%s)
( timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.stderr"; echo "bench rc=$? in $(( $(date +%s) - T0 )) s" )
python - "$OUT/bench_driver_cmd.json" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print("value %.4e  ms %.3f  frac %.4f traffic/alg %.3f  cpu %.3e (%s cores)  phases %s" % (o["value"], o["ms_per_step"], o["roofline"]["frac"],
      (o["roofline"]["traffic"] or 0) / o["roofline"]["algorithmic_bytes_per_launch"], o["cpu_baseline"]["value"] or 0, o["cpu_baseline"]["cores"], o["phases_s"]))
print("check", o["check"]["ok"], "secondary", len(o["secondary"]), "errors", [e["workload"][:40] for e in o["secondary"] if "error" in e])
g = [e for e in o["secondary"] if e["workload"].startswith("GAT")][0]
print("GAT", g["ms"], g["frac_whole_step"], g.get("frac_with_gemm_operands"), g.get("gemm_tflops_over_the_whole_pass"))
PY
