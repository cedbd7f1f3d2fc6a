# round 6, GPU job 3: where the time outside the row kernels goes in one rank's share of config 4 (2.5D dense-replicate), of the 2.5D
# sparse-replicating schedule and of 15d_fusion1 — kernel + memory-copy timelines of the solo replay; chunk shapes of the fusion1 mesh on paced links
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_job3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
trace() {  # $1 = tag, rest = arguments of rank_share_schedule.py
  tag=$1; shift
  python "$R/tools/rank_share_schedule.py" "$@" 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee "$OUT/rank_share_$tag.log"
  rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/prof_$tag" -o t -- python "$R/tools/rank_share_schedule.py" "$@" > /dev/null 2>&1
  DB=$(find "$OUT/prof_$tag" -name "*_results.db" | head -1)
  python "$R/tools/rocpd_timeline.py" "$DB" --busy --last-ms "$LAST" --min-us 5 > "$OUT/timeline_$tag.txt" 2>&1
  head -3 "$OUT/timeline_$tag.txt"
  rm -rf "$OUT/prof_$tag"
}
LAST=19 trace cfg4_25d_dense --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256
LAST=10 trace cfg4_25d_dense_sddmm --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256 --op sddmm
LAST=10 trace cfg4_25d_dense_spmm --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256 --op spmm
LAST=22 trace 25d_sparse --alg 25d_sparse_replicate --p 8 --c 2 --kind er --logm 20 --ef 96 --r 128
LAST=25 trace 15d_fusion1 --alg 15d_fusion1 --p 8 --c 1 --kind er --logm 20 --ef 96 --r 128
for T in "1,2,2,2,1,1" "1,2,1" "1,1"; do
  echo "== HNH_MESH_TAPER=$T" | tee -a "$OUT/overlap_accumulator_p8_chunk_shapes.log"
  HNH_MESH_TAPER=$T python "$R/tools/overlap_probe_accumulator.py" --p 8 --op fused --gbps 60,100 2>&1 | grep -v "amdgpu.ids\|R-mat" | tee -a "$OUT/overlap_accumulator_p8_chunk_shapes.log"
done
