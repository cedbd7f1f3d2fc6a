#!/bin/bash
# rocprofv3 evidence for ONE rank's share of BASELINE config 3 (tools/rank_share_probe.py): kernel-trace stats and the
# FETCH_SIZE / WRITE_SIZE passes (each --pmc set in its own run, never combined with a trace domain).
# Exports land in gpurun_out/profiles_export/ (copy them into profiles/).  Usage: tools/profile_rank_share.sh TAG [CHUNKS]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_rank_share}
Q=${2:-4}
OUT=$R/gpurun_out/profiles_export
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/rank_share_probe.py --chunks $Q --iters 5"
$CMD 2>&1 | grep "rank 0" | tee "$OUT/${TAG}_q${Q}.log"
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prs_stats" -o stats -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$R/gpurun_out/prs_fetch" -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$R/gpurun_out/prs_write" -o write -- $CMD > /dev/null 2>&1
S=$(find "$R/gpurun_out/prs_stats" -name "*_results.db" | head -1)
F=$(find "$R/gpurun_out/prs_fetch" -name "*_results.db" | head -1)
W=$(find "$R/gpurun_out/prs_write" -name "*_results.db" | head -1)
python "$R/tools/rocpd_summary.py" --tag "${TAG}_q${Q}" --stats "$S" --pmc "$F" "$W" --workload-key "rank_share_p8_q${Q}" --out "$OUT/rank_share_tmp" | tee -a "$OUT/${TAG}_q${Q}.log"
cp "$OUT/rank_share_tmp/${TAG}_q${Q}_kernel_stats.csv" "$OUT/" 2>/dev/null
rm -rf "$OUT/rank_share_tmp" "$R/gpurun_out/prs_stats" "$R/gpurun_out/prs_fetch" "$R/gpurun_out/prs_write"
