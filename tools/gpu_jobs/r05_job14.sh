# round 5, GPU job 14: the input side at size, with its timings printed
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job14
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k input_side > "$OUT/input_side_at_size.log" 2>&1; echo rc=$? >> "$OUT/input_side_at_size.log" )
grep -a "input side at size\|passed\|failed\|rc=" "$OUT/input_side_at_size.log"
