# round 4, final code: kernel + schedule GPU tests (with the folded SDDMM / borrowed value arrays), the whole sddmmA two ways, the
# headline bench line with its rocprofv3 kernel-trace stats (same command), the C++ drop-in driver without any environment in the
# same run, and the kernel list of steady-state calls on the R-MAT graph (no structure kernels after the first call).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final
mkdir -p "$OUT"
python -c "import torch" 2>/dev/null
( cd "$R" && timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_schedules_gpu.py -x -q > "$OUT/gputests_kernels_schedules.log" 2>&1; echo rc=$? >> "$OUT/gputests_kernels_schedules.log" )
tail -n 6 "$OUT/gputests_kernels_schedules.log"
( cd "$R" && timeout 300 python tools/kbench.py --r 16,128 --ops fold > "$OUT/kbench_fold.log" 2>&1 )
cat "$OUT/kbench_fold.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"
tail -c 400 "$OUT/bench_n1.json"
( cd "$R/examples" && for i in 1 2; do timeout 120 ./bench_er 20 96 15d_fusion2 128 1 "$OUT/bench_er_cpp_driver_$i.json" fused > "$OUT/bench_er_$i.log" 2>&1; done; tail -n 3 "$OUT/bench_er_2.log" )
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_stats" -o stats -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-check --no-secondary > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_stats" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag r04_final_cfg2 --stats "$S" --out "$OUT"
rm -rf "$R/gpurun_out/prof_stats"
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rmat" -o stats -- python "$R/bench.py" --workload rmat --edge-factor 44 --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-check --no-secondary > "$OUT/bench_rmat.json" 2>/dev/null
S=$(find "$R/gpurun_out/prof_rmat" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag r04_final_rmat --stats "$S" --kernel-like "::row_kernel<" --out "$OUT"
rm -rf "$R/gpurun_out/prof_rmat"
ls -la "$OUT"
