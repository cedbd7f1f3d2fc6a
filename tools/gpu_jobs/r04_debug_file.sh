mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import hnh_testlib as T
T.write_symmetric_mtx_with_duplicates("/tmp/g.mtx", 500, 3)
PY
export HNH_PERMUTE_SEED=5
for variant in "default" "cus0"; do
  for r in 64 32; do
    echo "=== $variant R=$r" >> gpurun_out/debug_file.log
    if [ "$variant" = "cus0" ]; then export HNH_COMM_CUS=0; else unset HNH_COMM_CUS; fi
    rm -f /tmp/out.json
    timeout 40 env AMD_LOG_LEVEL=3 examples/bench_file /tmp/g.mtx 15d $r 1 /tmp/out.json vanilla > /tmp/o.log 2> /tmp/e.log
    echo "rc=$?" >> gpurun_out/debug_file.log
    tail -3 /tmp/o.log >> gpurun_out/debug_file.log
    grep -v "hipGetDevice\|hipSetDevice" /tmp/e.log | tail -25 | cut -c1-260 >> gpurun_out/debug_file.log
  done
done
cat gpurun_out/debug_file.log | tail -150
