# round 5, GPU job 12: SIGTERM to `python bench.py --gpus 2` on real HIP processes (two ranks share this GPU): the launcher must forward rank 0's line in hand
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_job12
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for DELAY in 16 9; do
  python "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --logm 18 > "$OUT/sigterm_after_${DELAY}s.json" 2> "$OUT/sigterm_after_${DELAY}s.stderr" &
  PID=$!
  sleep $DELAY
  kill -TERM $PID
  wait $PID
  echo "delay $DELAY: launcher rc=$? lines=$(wc -l < "$OUT/sigterm_after_${DELAY}s.json")"
  python - "$OUT/sigterm_after_${DELAY}s.json" <<'PY'
import json, sys
txt = open(sys.argv[1]).read().strip()
o = json.loads(txt.splitlines()[-1]) if txt else {}
print({k: o.get(k) for k in ("value", "ms_per_step", "incomplete", "error", "exit_codes")}, (o.get("phases_s") or {}))
PY
done
