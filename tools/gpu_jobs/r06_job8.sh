# round 6, GPU job 8 (A/B): sddmmA / spmmA / fusedSpMM at config 2's size through the operator, the tree of the round's first commit against HEAD
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job8
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for k in 1 2; do
  python "$R/tmp_ab_old/probe.py" "$R/tmp_ab_old" 2>&1 | grep "R=" | tee -a "$OUT/ab.log"
  python "$R/tmp_ab_old/probe.py" "$R" 2>&1 | grep "R=" | tee -a "$OUT/ab.log"
done
