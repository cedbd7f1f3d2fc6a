# round 6, GPU job 12: test_bench_processes_share_one_gpu[4] repeated (one run of the long suite ended with an "incomplete" line)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job12
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
for k in 1 2 3 4 5 6 7 8; do
  HNH_LONG_TESTS=1 timeout 600 python -m pytest tests/test_multigpu_gpu.py -m gpu -q -x -k "processes_share_one_gpu" 2>&1 | tail -25 > "$OUT/run_$k.log"
  tail -1 "$OUT/run_$k.log"
  grep -l "failed" "$OUT/run_$k.log" > /dev/null && head -40 "$OUT/run_$k.log"
done
