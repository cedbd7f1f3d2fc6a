#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/mains/{bench_erdos_renyi,bench_file,bench_heatmap,scratch}: the reference's own
# main()s (+ benchmark_dist.cpp, the harness they call), UNMODIFIED, compiled against THIS repository's class headers through
# include/compat and linked with lib/libhnh_host.so — the source-level drop-in, demonstrated with the reference's own drivers.
# The sources are compiled from a temp dir that is deleted afterwards (their quoted #includes would otherwise find the reference's
# headers next to them); nothing from the reference enters the repository.  Outputs go only to oracle/_ref/ (git-ignored; travels
# to the GPU box, where tests/test_zz_reference_mains_gpu.py runs them on the HIP library).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
REF="${REF:-/root/reference}"
OUT="$HERE/_ref/mains"
if [ ! -d "$REF" ]; then echo "build_ref_mains: $REF absent (GPU box?) - keeping prebuilt $OUT"; exit 0; fi
mkdir -p "$OUT"
TMP="$(mktemp -d /tmp/hnh_ref_mains.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
for f in benchmark_dist.cpp benchmark_dist.hpp bench_erdos_renyi.cpp bench_file.cpp bench_heatmap.cpp scratch.cpp; do cp "$REF/$f" "$TMP/"; done
FLAGS="-O2 -std=c++17 -fopenmp -w -I$ROOT/include/compat -I$ROOT/distributed_sddmm_amd/csrc/host -I$ROOT/include"
LINK="-L$ROOT/distributed_sddmm_amd/lib -lhnh_host -ldl -lpthread -Wl,-rpath,\$ORIGIN/../../../distributed_sddmm_amd/lib"
g++ $FLAGS -c "$TMP/benchmark_dist.cpp" -o "$TMP/benchmark_dist.o"
for m in bench_erdos_renyi bench_file bench_heatmap; do g++ $FLAGS "$TMP/benchmark_dist.o" "$TMP/$m.cpp" -o "$OUT/$m" $LINK & done
g++ $FLAGS "$TMP/scratch.cpp" -o "$OUT/scratch" $LINK &
wait
for m in bench_erdos_renyi bench_file bench_heatmap scratch; do test -x "$OUT/$m"; done
echo "build_ref_mains: ok -> $OUT"
