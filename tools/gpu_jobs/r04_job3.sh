mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 900 python tools/kbench.py --r 8,16,32,64,128,256 --ops plan,coo > gpurun_out/kbench_c.log 2>&1
timeout 600 python tools/kbench.py --rmat --ef 44 --r 128 --ops plan >> gpurun_out/kbench_c.log 2>&1
cat gpurun_out/kbench_c.log
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_schedules_gpu.py -x -q > gpurun_out/gputests_c.log 2>&1; echo rc=$? >> gpurun_out/gputests_c.log )
tail -8 gpurun_out/gputests_c.log
