# round 6, GPU job 10: the long variants of the GPU tests (HNH_LONG_TESTS=1: 4-process bench, larger sweeps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job10
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( time HNH_LONG_TESTS=1 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > "$OUT/gputests_long.log" 2>&1
tail -6 "$OUT/gputests_long.log"
