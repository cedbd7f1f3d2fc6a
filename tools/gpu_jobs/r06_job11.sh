# round 6, GPU job 11: the 4-process small bench of test_bench_processes_share_one_gpu[4], to read why its line is marked incomplete
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job11
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
for k in 1 2; do
( time python3 bench.py --gpus 4 --steps 2 --warmup 1 --logm 14 --edge-factor 16 --r 32 --no-cpu-baseline --probe-timeout 240 > "$OUT/bench4_$k.json" 2> "$OUT/bench4_$k.stderr" ) 2> "$OUT/bench4_$k.time"
cp bench_secondary.json "$OUT/bench4_${k}_full.json"
python3 -c "
import json
d=json.load(open('$OUT/bench4_${k}_full.json'))
print('incomplete:', d.get('incomplete'))
print('failures:', d['config'].get('route_tuning_failures'))
print('stops:', d['config'].get('budget_stops'))
for k_,v in d['config'].get('route_tuning_ms_per_step',{}).items(): print(v, k_)
print(d.get('phases_s'))
"
grep -v "amdgpu.ids" "$OUT/bench4_$k.stderr" | tail -15
done
