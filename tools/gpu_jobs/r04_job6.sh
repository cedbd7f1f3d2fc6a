mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 600 python tools/overlap_probe_accumulator.py --p 4 --gbps 0,40,60,100,150 > gpurun_out/overlap_acc_fusion1.log 2>&1
timeout 600 python tools/overlap_probe_accumulator.py --p 4 --alg 25d_dense_replicate --gbps 0,40,60,100,150 > gpurun_out/overlap_acc_25d.log 2>&1
timeout 600 python tools/overlap_probe_accumulator.py --p 2 --gbps 0,60,100 > gpurun_out/overlap_acc_fusion1_p2.log 2>&1
tail -n 8 gpurun_out/overlap_acc_fusion1.log; tail -n 8 gpurun_out/overlap_acc_25d.log; tail -n 6 gpurun_out/overlap_acc_fusion1_p2.log
