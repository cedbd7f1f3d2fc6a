# round 6, GPU job 5: wide operands in 128-column slabs (un-fused SDDMM / SpMM at R = 256 / 384 / 512 / 640, slabs on / off), their parity tests,
# and the profiled kernel times behind cross-stream waits (the tiny dispatch in front of every start event)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job5
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 ) | tee "$OUT/gputests_kernels.log"
cd /tmp && export TMPDIR=/tmp
for S in 0 1; do
  echo "== HNH_WIDE_SLABS=$S" | tee -a "$OUT/kbench_wide_slabs.log"
  HNH_WIDE_SLABS=$S python "$R/tools/kbench.py" --r 384,512,640 --ops plan --iters 5 2>&1 | grep -v "amdgpu.ids\|generated" | tee -a "$OUT/kbench_wide_slabs.log"
done
echo "== HNH_WIDE_SLABS=1 HNH_SLAB_MIN_R=256 (R = 256: two slabs against four panels)" | tee -a "$OUT/kbench_wide_slabs.log"
HNH_SLAB_MIN_R=256 python "$R/tools/kbench.py" --r 256 --ops plan --iters 5 2>&1 | grep -v "amdgpu.ids\|generated" | tee -a "$OUT/kbench_wide_slabs.log"
HNH_WIDE_SLABS=0 python "$R/tools/kbench.py" --r 256 --ops plan --iters 5 2>&1 | grep -v "amdgpu.ids\|generated" | tee -a "$OUT/kbench_wide_slabs.log"
python "$R/tools/fusion1_probe.py" --chunks "1,2,1" 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee "$OUT/fusion1_rank_share_tick.log"
