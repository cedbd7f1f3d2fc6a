mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import hnh_testlib as T
T.write_symmetric_mtx_with_duplicates("/tmp/g.mtx", 500, 3)
PY
export HNH_PERMUTE_SEED=5
L=gpurun_out/debug_file3.log
: > $L
hangs=0
for i in $(seq 1 60); do
  alg=15d_fusion2; [ $((i % 2)) -eq 0 ] && alg=25d
  rm -f /tmp/out.json
  LD_PRELOAD=$PWD/tools/debug/libbt.so AMD_LOG_LEVEL=3 examples/bench_file /tmp/g.mtx $alg 64 1 /tmp/out.json vanilla > /tmp/o.log 2> /tmp/e.log &
  pid=$!
  for t in $(seq 1 80); do sleep 0.1; kill -0 $pid 2>/dev/null || break; done
  if kill -0 $pid 2>/dev/null; then
    hangs=$((hangs+1))
    echo "=== iter $i $alg HUNG" >> $L
    for tid in $(ls /proc/$pid/task); do echo "tid $tid wchan=$(cat /proc/$pid/task/$tid/wchan 2>/dev/null) $(grep -m1 State /proc/$pid/task/$tid/status)" >> $L; done
    kill -USR1 $pid; sleep 0.5
    grep -v "hipGetDevice\|hipSetDevice" /tmp/e.log | tail -40 | cut -c1-300 >> $L
    kill -9 $pid
    [ $hangs -ge 3 ] && break
  fi
  wait $pid 2>/dev/null
done
echo "iterations $i hangs $hangs" >> $L
tail -150 $L
