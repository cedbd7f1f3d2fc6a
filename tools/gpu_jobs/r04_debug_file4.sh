mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import hnh_testlib as T
T.write_symmetric_mtx_with_duplicates("/tmp/g.mtx", 500, 3)
PY
export HNH_PERMUTE_SEED=5
L=gpurun_out/debug_file4.log
: > $L
run_variant() {
  name=$1; shift
  hangs=0; n=0
  for i in $(seq 1 80); do
    alg=15d_fusion2; [ $((i % 2)) -eq 0 ] && alg=25d
    rm -f /tmp/out.json
    env "$@" LD_PRELOAD=$PWD/tools/debug/libbt.so examples/bench_file /tmp/g.mtx $alg 64 1 /tmp/out.json vanilla > /tmp/o.log 2> /tmp/e.log &
    pid=$!
    for t in $(seq 1 60); do sleep 0.1; kill -0 $pid 2>/dev/null || break; done
    n=$((n+1))
    if kill -0 $pid 2>/dev/null; then
      hangs=$((hangs+1))
      if [ $hangs -le 0 ]; then
        echo "=== $name iter $i $alg HUNG" >> $L
        for tid in $(ls /proc/$pid/task); do echo "tid $tid wchan=$(cat /proc/$pid/task/$tid/wchan 2>/dev/null) $(grep -m1 State /proc/$pid/task/$tid/status)" >> $L; done
        kill -USR1 $pid; sleep 0.5
        tail -40 /tmp/e.log | cut -c1-300 >> $L
      fi
      kill -9 $pid
    fi
    wait $pid 2>/dev/null
  done
  echo "variant $name: $hangs hangs in $n runs" >> $L
}
run_variant masked X=1
run_variant nomask HNH_COMM_CUS=0
run_variant masked_omp8 OMP_NUM_THREADS=8
run_variant masked_passive OMP_WAIT_POLICY=passive
tail -120 $L
