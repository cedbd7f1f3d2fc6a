#!/bin/bash
# Counter passes over a command, means PER KERNEL NAME (every kernel the command launches).  Each --pmc set in its own rocprofv3
# run, never combined with a trace domain.  Usage:  tools/pmc_by_kernel.sh TAG -- <command>
# Writes gpurun_out/profiles_export/${TAG}_pmc_by_kernel.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift 2
OUT=$R/gpurun_out/profiles_export
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "TA_TA_BUSY_sum TD_TD_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"
)
# PMC_SETS="1 2": only those sets (indices into SETS)
if [ -n "${PMC_SETS:-}" ]; then
  PICK=()
  for k in $PMC_SETS; do PICK+=("${SETS[$k]}"); done
  SETS=("${PICK[@]}")
fi
DBS=()
i=0
for s in "${SETS[@]}"; do
  d=$R/gpurun_out/pmcK_${TAG}_$i
  rm -rf "$d"
  # shellcheck disable=SC2086
  timeout 600 rocprofv3 --pmc $s -d "$d" -o p -- "$@" > "$OUT/${TAG}_passK$i.log" 2>&1 || echo "pass $i ($s) failed"
  db=$(find "$d" -name "*_results.db" | head -1)
  [ -n "$db" ] && DBS+=("$db")
  i=$((i+1))
done
python - "${DBS[@]}" > "$OUT/${TAG}_pmc_by_kernel.txt" <<'PY'
import sqlite3, sys, collections
acc = collections.defaultdict(list)
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    for k, c, v, d in cur.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        acc[(k, c)].append((v, d))
names = sorted({k for k, _ in acc})
for k in names:
    ctrs = sorted(c for kk, c in acc if kk == k)
    n = len(acc[(k, ctrs[0])])
    dur = sum(d for _, d in acc[(k, ctrs[0])]) / n
    if dur < 2e5:  # skip kernels shorter than 0.2 ms
        continue
    print("== %s  (%d dispatches, mean %.3f ms under the counters)" % (k[:150], n, dur / 1e6))
    for c in ctrs:
        vals = [v for v, _ in acc[(k, c)]]
        print("   %-40s %.4e" % (c, sum(vals) / len(vals)))
PY
for j in $(seq 0 $((i-1))); do rm -rf "$R/gpurun_out/pmcK_${TAG}_$j"; done
cat "$OUT/${TAG}_pmc_by_kernel.txt"
