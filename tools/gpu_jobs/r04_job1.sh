mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_a.log 2>&1; echo rc=$? >> gpurun_out/gputests_a.log )
tail -5 gpurun_out/gputests_a.log
# headline: old bench with its CU search (0 = no mask, 16 = the new default), and the C++ driver with no environment
timeout 600 python bench.py --no-cpu-baseline --no-live-traffic > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; tail -c 1500 gpurun_out/bench_a.json
rm -f /tmp/er.json; timeout 600 examples/bench_er 20 96 15d_fusion2 128 1 /tmp/er.json fused > gpurun_out/bench_er_a.log 2>&1; cat /tmp/er.json | head -40 > gpurun_out/bench_er_a.json
# kernel level: R sweep with hints, and the R-MAT hub path
timeout 900 python tools/kbench.py --r 8,16,32,64,128,256 --panels --ops coo > gpurun_out/kbench_a.log 2>&1
timeout 600 python tools/kbench.py --rmat --ef 44 --r 128 --panels --ops fused >> gpurun_out/kbench_a.log 2>&1
tail -40 gpurun_out/kbench_a.log
