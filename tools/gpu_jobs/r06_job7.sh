# round 6, GPU job 7: counter traffic of the wide un-fused passes, one wide pass against 128-column slabs (R = 512, config 2's matrix):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, summed per kernel instance over the run of tools/kbench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job7
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for S in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_${S}_$C
    HNH_WIDE_SLABS=$S timeout 900 rocprofv3 --pmc $C -d "$d" -o p -- python "$R/tools/kbench.py" --r 512 --ops plan --iters 3 > "$OUT/kbench_slabs${S}_$C.log" 2>&1
  done
  python3 - "$OUT" "$S" <<'PY' | tee -a "$OUT/wide_slabs_counter_traffic.txt"
import glob, sqlite3, sys, collections
out, s = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob("%s/pmc_%s_%s/**/*_results.db" % (out, s, c), recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, name, v in cur.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%row_kernel%'"):
            short = k.split("row_kernel<")[1].split(">")[0] if "row_kernel<" in k else k[:40]
            a = acc[short][name]
            a[0] += 1
            a[1] += v
print("== HNH_WIDE_SLABS=%s  (kbench --r 512 --ops plan --iters 3: every op runs 1 warm-up + 3 timed calls = 4 calls; KB summed over all its dispatches)" % s)
for short, d in sorted(acc.items()):
    f, w = d.get("FETCH_SIZE", [0, 0.0]), d.get("WRITE_SIZE", [0, 0.0])
    # gfx950: FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane reads at 64 bytes (x 2); WRITE_SIZE as reported
    print("row_kernel<%s>: %4d dispatches, fetch %.3f GB (corrected x2: %.3f GB), write %.3f GB" % (short, f[0], f[1] * 1024 / 1e9, 2 * f[1] * 1024 / 1e9, w[1] * 1024 / 1e9))
PY
  rm -rf "$OUT"/pmc_${S}_*
done
grep -h "R=512" "$OUT"/kbench_slabs*_FETCH_SIZE.log | tee -a "$OUT/wide_slabs_counter_traffic.txt"
