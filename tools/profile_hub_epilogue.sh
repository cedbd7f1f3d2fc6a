#!/bin/bash
# Evidence for the row epilogue of blocks WITH hub rows (R-MAT 2^20): ALS-CG and the GAT head on a skewed graph, timed, and the
# kernel list of each (rocprofv3 --kernel-trace --stats): no row_epilogue_kernel launch may appear — short rows get their
# epilogue in row_kernel<kFusedCg>, hub rows in reduce_long_kernel.  Usage: tools/profile_hub_epilogue.sh TAG
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_hub_epilogue}
OUT=$R/gpurun_out/profiles_export
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HNH_PROFILE_RMAT_EDGES=${HNH_PROFILE_RMAT_EDGES:-50000000}
python "$R/tools/als_profile.py" 20 2>&1 | grep -v "R-mat gen" | tee "$OUT/${TAG}_als.log"
HNH_PROFILE_RMAT_EDGES=8000000 python "$R/tools/gat_profile.py" 18 15d_fusion2 2>&1 | grep -v "R-mat gen" | tee "$OUT/${TAG}_gat.log"
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/hub_als" -o s -- python "$R/tools/als_profile.py" 20 > /dev/null 2>&1
HNH_PROFILE_RMAT_EDGES=8000000 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/hub_gat" -o s -- python "$R/tools/gat_profile.py" 18 15d_fusion2 > /dev/null 2>&1
for k in als gat; do
  db=$(find "$R/gpurun_out/hub_$k" -name "*_results.db" | head -1)
  python "$R/tools/rocpd_summary.py" --tag "${TAG}_$k" --stats "$db" --out "$OUT" > /dev/null
  echo "--- kernels of the $k run (name, calls, avg us):" | tee -a "$OUT/${TAG}_$k.log"
  python - "$OUT/${TAG}_${k}_kernel_stats.csv" <<'PY' | tee -a "$OUT/${TAG}_$k.log"
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:16]:
    print("   %-110s calls %6s avg %10.1f us" % (r[0][:110], r[1], float(r[3])))
rows = list(csv.reader(open(sys.argv[1])))[1:]
print("   row_epilogue_kernel launches: %d" % sum(int(r[1]) for r in rows if "row_epilogue_kernel" in r[0]))
PY
  rm -rf "$R/gpurun_out/hub_$k"
done
