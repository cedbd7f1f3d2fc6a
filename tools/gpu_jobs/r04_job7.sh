mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
GPU_MAX_HW_QUEUES=32 timeout 600 python tools/loopback_bench.py --p 8 --alg 15d_sparse --r 128 > gpurun_out/loopback_p8_15d_sparse.log 2>&1
tail -n 6 gpurun_out/loopback_p8_15d_sparse.log
