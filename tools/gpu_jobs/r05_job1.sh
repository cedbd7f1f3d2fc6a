# round 5, GPU job 1: first numbers for what the round-4 review asked to see
#   (1) the default bench line with the new secondary entries (one rank's share of configs 3 / 4 / 5, config 1 as typed) and phases_s
#   (2) multi-process transport tests (ipc-pull changes: eviction between groups, unaligned pulls)
#   (3) narrow accumulating SDDMM baseline (row kernel vs COO kernel at R = 8 / 16)
#   (4) wide operands: panels x Out overwrite at R = 256 / 384 / 512 with counter traffic
#   (5) counter traffic of one rank's share of config 3 at the default chunk shape and at one chunk
# usage: gpurun --timeout 1500 -- bash tools/gpu_jobs/r05_job1.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job1
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
( timeout 500 python "$R/bench.py" --no-cpu-baseline > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"; echo "bench rc=$?" )
tail -c 1500 "$OUT/bench_n1.json"
cd "$R"
( timeout 400 python -m pytest tests/test_multigpu_gpu.py -x -q --durations=5 > "$OUT/gputests_multigpu.log" 2>&1; echo rc=$? >> "$OUT/gputests_multigpu.log" )
tail -n 8 "$OUT/gputests_multigpu.log"
cd /tmp
( timeout 200 python "$R/tools/kbench.py" --r 8,16,32 --ops plan,coo --iters 5 > "$OUT/kbench_narrow_baseline.log" 2>&1 )
cat "$OUT/kbench_narrow_baseline.log"
( timeout 600 python "$R/tools/wide_panels.py" --traffic --json "$OUT/wide_panels.json" > "$OUT/wide_panels.log" 2>&1 )
cat "$OUT/wide_panels.log"
for Q in default 1; do
  if [ "$Q" = default ]; then CH=""; else CH="--chunks $Q"; fi
  CMD="python $R/tools/rank_share_probe.py $CH --iters 5"
  $CMD 2>&1 | grep "rank 0" | tee "$OUT/rank_share_q$Q.log"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$R/gpurun_out/prs_fetch" -o fetch -- $CMD > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$R/gpurun_out/prs_write" -o write -- $CMD > /dev/null 2>&1
  F=$(find "$R/gpurun_out/prs_fetch" -name "*_results.db" | head -1)
  W=$(find "$R/gpurun_out/prs_write" -name "*_results.db" | head -1)
  python "$R/tools/rocpd_summary.py" --tag "r05_rank_share_q$Q" --pmc "$F" "$W" --workload-key "rank_share_p8_q$Q" --out "$OUT/rs_tmp" | tee -a "$OUT/rank_share_q$Q.log"
  rm -rf "$OUT/rs_tmp" "$R/gpurun_out/prs_fetch" "$R/gpurun_out/prs_write"
done
ls -la "$OUT"
