# round 6, GPU job 6: the state of the tree as the driver will see it — smoke, all GPU tests, the driver's bench command, kernel trace of it
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${JOBTAG:-r06_job6}
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > "$OUT/smoke.log" 2>&1
tail -3 "$OUT/smoke.log"
( time python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > "$OUT/gputests.log" 2>&1
tail -6 "$OUT/gputests.log"
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.stderr" ) 2> "$OUT/bench_driver_cmd.time"
cp bench_secondary.json "$OUT/bench_driver_cmd_full_record.json" 2>/dev/null
python3 -c "
import json
l=open('$OUT/bench_driver_cmd.json').read().strip().splitlines()[-1]; d=json.loads(l)
print('bytes', len(l)+1, 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'phases', d['phases_s'])
for r in d['secondary']: print(r)
print(d['cpu_baseline'])
" | tee "$OUT/bench_line_summary.txt"
wc -c "$OUT/bench_driver_cmd.stderr"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o cfg2 -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-live-traffic > "$OUT/bench_under_rocprof.json" 2> /dev/null
DB=$(find "$OUT/prof" -name "*_results.db" | head -1)
python3 "$R/tools/rocpd_summary.py" --tag r06_final_cfg2 --stats "$DB" --out "$OUT" 2>&1 | tee "$OUT/cfg2_kernel_stats.txt"
rm -rf "$OUT/prof"
python "$R/tools/kbench.py" --rmat --logm 20 --ef 44 --r 128 --ops plan --iters 5 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee "$OUT/kbench_rmat.log"
