mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
( timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo rc=$? >> gpurun_out/bench_b.err )
tail -c 3000 gpurun_out/bench_b.json; tail -5 gpurun_out/bench_b.err
( timeout 1700 python -m pytest tests/test_multigpu_gpu.py tests/test_schedules_gpu.py -x -q -k "bench_processes or rccl_only or cpp_dropin" > gpurun_out/gputests_b.log 2>&1; echo rc=$? >> gpurun_out/gputests_b.log )
tail -30 gpurun_out/gputests_b.log
