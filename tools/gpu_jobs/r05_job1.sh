# round 5, first GPU job: everything round 4 added after its GPU budget was spent and therefore has not run on a GPU yet —
#   (1) the new GPU tests on their own, first (the reference's unmodified mains on the HIP library, incl. the exit-handler teardown of the
#       process world; examples/c_operator inside test_cpp_dropin_driver; the golden branches of the full-size tests),
#   (2) the whole GPU suite with durations (what the driver runs at the round's end),
#   (3) the default bench line (new secondary entry: the reference's printed weak-scaling point) and its rocprofv3 kernel stats.
# usage: gpurun --timeout 1500 -- bash tools/gpu_jobs/r05_job1.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job1
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 300 python -m pytest tests/test_zz_reference_mains_gpu.py -x -q --durations=5 > "$OUT/gputests_reference_mains.log" 2>&1; echo rc=$? >> "$OUT/gputests_reference_mains.log" )
tail -n 12 "$OUT/gputests_reference_mains.log"
( timeout 300 python -m pytest tests/test_schedules_gpu.py -x -q -k "cpp_dropin or at_scale" > "$OUT/gputests_dropin.log" 2>&1; echo rc=$? >> "$OUT/gputests_dropin.log" )
tail -n 5 "$OUT/gputests_dropin.log"
( timeout 400 python -m pytest tests/test_fullsize_gpu.py -x -q --durations=8 > "$OUT/gputests_fullsize.log" 2>&1; echo rc=$? >> "$OUT/gputests_fullsize.log" )
tail -n 14 "$OUT/gputests_fullsize.log"
( timeout 900 python -m pytest tests/ -x -q -m gpu --durations=15 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 24 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
timeout 400 python "$R/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"
tail -c 600 "$OUT/bench_n1.json"
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_stats" -o stats -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-check --no-secondary > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_stats" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag r05_job1_cfg2 --stats "$S" --out "$OUT"
rm -rf "$R/gpurun_out/prof_stats"
ls -la "$OUT"
