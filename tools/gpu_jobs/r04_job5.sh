mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_acc
rocprofv3 --kernel-trace -d /tmp/prof_acc -o acc -- python $R/tools/overlap_probe_accumulator.py --p 4 --gbps 60 --calls 1 > $R/gpurun_out/overlap_acc_traced.log 2>&1
DB=$(find /tmp/prof_acc -name "*_results.db" | head -1)
python $R/tools/rocpd_timeline.py $DB --last-ms 45 --min-us 100 > $R/gpurun_out/overlap_acc_timeline.txt 2>&1
tail -5 $R/gpurun_out/overlap_acc_traced.log
head -120 $R/gpurun_out/overlap_acc_timeline.txt
