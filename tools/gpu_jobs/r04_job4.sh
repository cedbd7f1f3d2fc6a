mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 600 python tools/overlap_probe_accumulator.py --p 4 --gbps 0,40,60,100 > gpurun_out/overlap_acc_fusion1.log 2>&1
timeout 600 python tools/overlap_probe_accumulator.py --p 4 --alg 25d_dense_replicate --gbps 0,40,60,100 > gpurun_out/overlap_acc_25d.log 2>&1
tail -8 gpurun_out/overlap_acc_fusion1.log gpurun_out/overlap_acc_25d.log
( timeout 1500 python bench.py --gpus 8 --logm 17 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_8proc.json 2> gpurun_out/bench_8proc.err; echo rc=$? >> gpurun_out/bench_8proc.err )
tail -c 2500 gpurun_out/bench_8proc.json; tail -5 gpurun_out/bench_8proc.err
