# round 4: the GEMM with 256 x 128 tiles of 8 waves (HNH_GEMM_WAVES=8) — parity, stand-alone rate and inside the GAT pipeline —
# then the whole GPU suite at the final code and the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_last
mkdir -p "$OUT"
cd "$R"
( timeout 200 python -m pytest tests/test_kernels_gpu.py -k "gemm" -x -q > "$OUT/gputests_gemm.log" 2>&1; echo rc=$? >> "$OUT/gputests_gemm.log" )
tail -n 3 "$OUT/gputests_gemm.log"
run() { echo "== $*" >> "$OUT/gat_gemm_waves.log"; ( env "$@" timeout 120 python tools/gat_profile.py 18 15d_fusion2 2>&1 | grep "GAT forward\|gemm_f64" >> "$OUT/gat_gemm_waves.log" ); }
run HNH_GEMM_WAVES=4
run HNH_GEMM_WAVES=8
run HNH_GEMM_WAVES=8 HNH_GAT_SERIAL=1
run HNH_GEMM_WAVES=4 HNH_GAT_SERIAL=1
cat "$OUT/gat_gemm_waves.log"
( timeout 640 python -m pytest tests/ -x -q -m gpu --durations=12 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 18 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
timeout 200 python "$R/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"
tail -c 600 "$OUT/bench_n1.json"
