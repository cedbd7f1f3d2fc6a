mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import hnh_testlib as T
T.write_symmetric_mtx_with_duplicates("/tmp/g.mtx", 500, 3)
PY
export HNH_PERMUTE_SEED=5
: > gpurun_out/debug_file2.log
for i in $(seq 1 25); do
  for alg in 15d 25d 15d_fusion2; do
    rm -f /tmp/out.json
    timeout 30 examples/bench_file /tmp/g.mtx $alg 64 1 /tmp/out.json vanilla > /tmp/o.log 2> /tmp/e.log
    rc=$?
    if [ $rc -ne 0 ]; then echo "iter $i $alg rc=$rc" >> gpurun_out/debug_file2.log; tail -3 /tmp/o.log >> gpurun_out/debug_file2.log; tail -5 /tmp/e.log >> gpurun_out/debug_file2.log; fi
  done
done
echo "loop done" >> gpurun_out/debug_file2.log
# the hang, if it shows: where is the process?
for i in 1 2 3; do timeout 300 python -m pytest tests/test_schedules_gpu.py -x -q -k cpp_dropin 2>&1 | tail -2 >> gpurun_out/debug_file2.log; done
cat gpurun_out/debug_file2.log
