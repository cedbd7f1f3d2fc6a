# round 6, GPU job 2: 15d_fusion1 on the mesh — GPU tests (schedules, kernels, multi-process), the rank's share ring vs mesh, the paced
# accumulator probe (ring / two halves / mesh reduce-scatter) for the SpMM, the SDDMM and the pair
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job2
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
export HNH_OBSERVED_LOG="$OUT/observed_errors.jsonl"
( time python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > "$OUT/gputests.log" 2>&1
tail -8 "$OUT/gputests.log"
unset HNH_OBSERVED_LOG
cd /tmp && export TMPDIR=/tmp
python "$R/tools/fusion1_probe.py" 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee "$OUT/fusion1_rank_share.log"
python "$R/tools/fusion1_probe.py" --p 4 --chunks "1,2,2,2,1,1" 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/fusion1_rank_share.log"
for OP in spmm sddmm fused; do
  python "$R/tools/overlap_probe_accumulator.py" --p 4 --op $OP --gbps 0,40,60,100 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/overlap_accumulator_p4.log"
done
python "$R/tools/overlap_probe_accumulator.py" --p 8 --op spmm --gbps 0,60,100 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/overlap_accumulator_p8.log"
