# round 5, GPU job 13: smoke + the whole GPU suite at the final commit
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job13
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee "$OUT/smoke.log"
( timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=5 > "$OUT/gputests_all.log" 2>&1; echo rc=$? >> "$OUT/gputests_all.log" )
tail -n 12 "$OUT/gputests_all.log"
