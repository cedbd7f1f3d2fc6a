# round 6, GPU job 9: eight bench processes on ONE GPU (ipc-pull between them): the N > 1 line has to say where its ranks ran
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job9
mkdir -p "$OUT"
cd "$R"
python -c "import torch" 2>/dev/null
( time python3 bench.py --gpus 8 --steps 5 --warmup 2 --budget-s 420 --no-cpu-baseline > "$OUT/bench_8proc_one_gpu.json" 2> "$OUT/bench_8proc_one_gpu.stderr" ) 2> "$OUT/bench_8proc_one_gpu.time"
cp bench_secondary.json "$OUT/bench_8proc_one_gpu_full_record.json" 2>/dev/null
python3 -c "
import json
l=open('$OUT/bench_8proc_one_gpu.json').read().strip().splitlines()[-1]; d=json.loads(l)
print('bytes', len(l)+1)
print(d['config']['workload'])
print('distinct_devices', d['config'].get('distinct_devices'), 'ranks', d['config'].get('ranks'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'frac_kernel', d['roofline']['frac_kernel'])
print({k: len(json.dumps(v)) for k, v in d.items()})
print(d.get('shed'), d['phases_s'])
" | tee "$OUT/bench_8proc_summary.txt"
tail -5 "$OUT/bench_8proc_one_gpu.stderr"
cat "$OUT/bench_8proc_one_gpu.time"
