# round 5, GPU job 8: L2 (TCC) request counters of the GAT pair's two kernels, each alone (counter collection serialises dispatches) — what
# they would have to share side by side; the small call after the empty counter phases were dropped.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job8
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
PMC_SETS="1" HNH_GAT_SERIAL=1 HNH_PROFILE_NO_GEMM=1 bash "$R/tools/pmc_by_kernel.sh" r05_gat_tcc -- python "$R/tools/gat_profile.py" 18 15d_fusion2 > "$OUT/gat_tcc_counters.txt" 2>&1
grep -A6 "gemm_f64_kernel\|row_kernel" "$OUT/gat_tcc_counters.txt" | cut -c1-200 | head -40
cp "$R/gpurun_out/profiles_export/r05_gat_tcc_pmc_by_kernel.txt" "$OUT/" 2>/dev/null
python "$R/tools/small_call_probe.py" 2>&1 | grep -v amdgpu.ids | tee "$OUT/small_call_after_trim.log"
python "$R/tools/small_call_probe.py" --iters 1000 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/small_call_after_trim.log"
