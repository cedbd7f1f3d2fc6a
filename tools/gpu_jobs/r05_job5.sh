# round 5, GPU job 5: the GEMM with prefetch distance two — its own tests, alone, and beside the attention pass (GAT pipeline vs serial)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job5
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_schedules_gpu.py -x -q -k "gemm or gat or window_groupings" > "$OUT/gputests_gemm_gat.log" 2>&1; echo rc=$? >> "$OUT/gputests_gemm_gat.log" )
tail -n 6 "$OUT/gputests_gemm_gat.log"
cd /tmp && export TMPDIR=/tmp
gat() {
  echo "== $*" | tee -a "$OUT/gat_prefetch2.log"
  env "$@" timeout 200 python "$R/tools/gat_profile.py" 18 15d_fusion2 2>&1 | grep -E "GAT forward|gemm_f64" | tee -a "$OUT/gat_prefetch2.log"
}
gat HNH_DUMMY=1
gat HNH_GAT_SERIAL=1
gat HNH_DUMMY=2
gat HNH_AUX_PRIORITY=high
gat HNH_GEMM_WAVES=8
ls -la "$OUT"
