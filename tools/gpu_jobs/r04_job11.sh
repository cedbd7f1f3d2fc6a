# round 4, GAT forward pass as a two-stage pipeline (product of head j + 1 on HNH_STREAM_AUX beside the attention pass of head j):
# parity tests, then serial vs pipelined on the benchmark layers (benchmark_dist.cpp:88-94) at 2^18 vertices, with the knobs that
# decide how the two kernels share a CU (row-kernel occupancy pad, stream priority), and the dispatch timeline of the default.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_gat
mkdir -p "$OUT"
cd "$R"
( timeout 400 python -m pytest tests/test_schedules_gpu.py -k "gat" tests/test_kernels_gpu.py -k "gat or gemm" -x -q > "$OUT/gputests_gat.log" 2>&1; echo rc=$? >> "$OUT/gputests_gat.log" )
tail -n 4 "$OUT/gputests_gat.log"
run() { echo "== $*" >> "$OUT/gat_ab.log"; ( env "$@" timeout 120 python tools/gat_profile.py 18 ${ALG:-15d_fusion2} 2>&1 | grep "GAT forward" >> "$OUT/gat_ab.log" ); }
run HNH_GAT_SERIAL=1
run HNH_DUMMY=1
run HNH_ROW_WAVES_CAP=0
run HNH_AUX_PRIORITY=low
run HNH_AUX_PRIORITY=high
run HNH_AUX_PRIORITY=low HNH_ROW_WAVES_CAP=0
run HNH_GAT_SERIAL=1
run HNH_DUMMY=1
ALG=15d_fusion1 run HNH_GAT_SERIAL=1
ALG=15d_fusion1 run HNH_DUMMY=1
cat "$OUT/gat_ab.log"
cd /tmp && export TMPDIR=/tmp
HNH_PROFILE_NO_GEMM=1 timeout 200 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_gat" -o gat -- python "$R/tools/gat_profile.py" 18 15d_fusion2 > /dev/null 2>&1
S=$(find "$R/gpurun_out/prof_gat" -name "*_results.db" | head -1)
python "$R/tools/rocpd_timeline.py" "$S" --last-ms 75 --min-us 50 > "$OUT/gat_pipeline_timeline.txt" 2>&1
head -n 60 "$OUT/gat_pipeline_timeline.txt"
rm -rf "$R/gpurun_out/prof_gat"
