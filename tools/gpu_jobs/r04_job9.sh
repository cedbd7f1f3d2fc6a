mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_tuples_gpu.py tests/test_multigpu_gpu.py -x -q --durations=15 > gpurun_out/gputests_e.log 2>&1; echo rc=$? >> gpurun_out/gputests_e.log )
tail -n 40 gpurun_out/gputests_e.log
