# round 5, GPU job 9: config 1's fused call with the performance counters' events off
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job9
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for V in 1 0; do
  echo "== HNH_PERF_COUNTERS=$V" | tee -a "$OUT/small_call_counters_off.log"
  HNH_PERF_COUNTERS=$V python "$R/tools/small_call_probe.py" --iters 1000 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/small_call_counters_off.log"
  HNH_PERF_COUNTERS=$V python "$R/tools/small_call_probe.py" --iters 1000 --alg 15d_fusion2 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/small_call_counters_off.log"
  HNH_PERF_COUNTERS=$V python "$R/tools/small_call_probe.py" --iters 1000 --p 1 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/small_call_counters_off.log"
done
