mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
( timeout 1700 python -m pytest tests/test_multigpu_gpu.py -x -q -k "torch_distributed_run" > gpurun_out/gputests_d.log 2>&1; echo rc=$? >> gpurun_out/gputests_d.log )
tail -n 25 gpurun_out/gputests_d.log
