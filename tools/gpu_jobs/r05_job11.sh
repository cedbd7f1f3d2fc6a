# round 5, GPU job 11: Infinity-Cache panels SMALLER than 512 MiB at the headline width (round 2 swept 512 MiB and up only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job11
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
( timeout 300 python "$R/tools/wide_panels.py" --r 128 --panels 1,2,3,4,6 --iters 7 2>&1 | grep -v amdgpu.ids > "$OUT/panels_r128.log" ); cat "$OUT/panels_r128.log"
( timeout 300 python "$R/tools/wide_panels.py" --r 64 --panels 1,2,3 --iters 7 2>&1 | grep -v amdgpu.ids > "$OUT/panels_r64.log" ); cat "$OUT/panels_r64.log"
( timeout 300 python "$R/tools/wide_panels.py" --r 128 --panels 2,3,4 --iters 7 2>&1 | grep -v amdgpu.ids >> "$OUT/panels_r128.log" ); tail -6 "$OUT/panels_r128.log"
