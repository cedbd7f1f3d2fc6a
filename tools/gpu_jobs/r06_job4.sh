# round 6, GPU job 4: the driver's bench command with the round's changes (15d_fusion1 on the mesh, stored SpMM outputs, shared phase events,
# profile leg without host syncs), the small-call probe (config 1 as typed, counters on / off), GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job4
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.stderr" ) 2> "$OUT/bench_driver_cmd.time"
cp bench_secondary.json "$OUT/bench_driver_cmd_full_record.json" 2>/dev/null
python3 -c "
import json,sys
l=open('$OUT/bench_driver_cmd.json').read().strip().splitlines()[-1]; d=json.loads(l)
print('bytes', len(l)+1, 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'phases', d['phases_s'])
for r in d['secondary']: print(r)
print(d['cpu_baseline'])
" | tee "$OUT/bench_line_summary.txt"
cd /tmp && export TMPDIR=/tmp
for V in 1 0; do
  echo "== HNH_PERF_COUNTERS=$V" | tee -a "$OUT/small_call.log"
  HNH_PERF_COUNTERS=$V python "$R/tools/small_call_probe.py" --iters 1000 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/small_call.log"
  HNH_PERF_COUNTERS=$V python "$R/tools/small_call_probe.py" --iters 1000 --p 1 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/small_call.log"
done
python "$R/tools/fusion1_probe.py" --chunks "1,2,1" 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee "$OUT/fusion1_rank_share.log"
GPU_MAX_HW_QUEUES=16 python "$R/tools/rank_share_schedule.py" --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee "$OUT/rank_share_cfg4_q16.log"
GPU_MAX_HW_QUEUES=32 python "$R/tools/rank_share_schedule.py" --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/rank_share_cfg4_q16.log"
cd "$R"
( time python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > "$OUT/gputests.log" 2>&1
tail -6 "$OUT/gputests.log"
