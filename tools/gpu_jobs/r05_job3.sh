# round 5, GPU job 3: (1) the GPU suite (input side at size fixed), (2) GAT pipeline co-residency by LDS budget — how many GEMM and row
# workgroups share a CU is decided by their LDS requests (GEMM 66 KiB static, row kernel's occupancy pad 32 KiB): try one GEMM workgroup +
# three row workgroups per CU; (3) panels 4 / 6 / 8 at R = 384, 512 on ONE box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job3
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=12 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 22 "$OUT/gputests_all.log"
cd /tmp && export TMPDIR=/tmp
gat() {  # label, env...
  echo "== $*" | tee -a "$OUT/gat_coresidency.log"
  env "$@" timeout 200 python "$R/tools/gat_profile.py" 18 15d_fusion2 2>&1 | grep -E "GAT forward|gemm_f64" | tee -a "$OUT/gat_coresidency.log"
}
gat HNH_DUMMY=1
gat HNH_GAT_SERIAL=1
gat HNH_GEMM_LDS_EXTRA=16384 HNH_ROW_WAVES_CAP=7
gat HNH_GEMM_LDS_EXTRA=16384 HNH_ROW_WAVES_CAP=5
gat HNH_GEMM_LDS_EXTRA=16384 HNH_ROW_WAVES_CAP=9
gat HNH_GEMM_LDS_EXTRA=0 HNH_ROW_WAVES_CAP=6
gat HNH_GEMM_LDS_EXTRA=0 HNH_ROW_WAVES_CAP=7
gat HNH_GEMM_LDS_EXTRA=32768 HNH_ROW_WAVES_CAP=10
gat HNH_GEMM_LDS_EXTRA=16384 HNH_ROW_WAVES_CAP=7 HNH_AUX_PRIORITY=high
gat HNH_GEMM_WAVES=8 HNH_ROW_WAVES_CAP=8
( timeout 400 python "$R/tools/wide_panels.py" --r 384,512 --panels 4,6,8 --json "$OUT/wide_panels_468.json" 2>&1 | grep -v amdgpu.ids > "$OUT/wide_panels_468.log" )
cat "$OUT/wide_panels_468.log"
ls -la "$OUT"
