#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/ref_driver{,_patched}: the reference's own hot-path
# sources (sparse_kernels.cpp, common.cpp + headers) compiled WHERE THEY LIE under $REF, against the
# three stand-in headers in oracle/stubs, linked with the real MKL 2021.4 (ILP64, gnu_thread) and
# MPICH 3.3.2 shared libraries from /opt/conda/lib (recipe: SURVEY.md Appendix B).  Nothing from the
# reference is copied into the repository: the one patched header needed for the 2.5D-dense variant
# (SpmatLocal.hpp:252 `else if` -> `if`) is generated in a temp dir that is
# deleted afterwards.  Outputs go only to oracle/_ref/ (git-ignored; travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${REF:-/root/reference}"
CONDA_LIB="${CONDA_LIB:-/opt/conda/lib}"
CONDA_INC="${CONDA_INC:-/opt/conda/include}"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then echo "build_ref: $REF absent (GPU box?) - keeping prebuilt $OUT"; exit 0; fi
mkdir -p "$OUT"
TMP="$(mktemp -d /tmp/hnh_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT

# private copies of the MPICH headers only (-I$CONDA_INC wholesale would shadow system headers)
mkdir -p "$TMP/inc" "$TMP/deps" "$TMP/patched"
for h in mpi.h mpicxx.h mpio.h mpif.h mpiof.h; do cp "$CONDA_INC/$h" "$TMP/inc/"; done
# link against symlinks so that /opt/conda/lib's old libstdc++ is never on a search path
for f in "$CONDA_LIB"/libmpi.so* "$CONDA_LIB"/libmkl_*.so* "$CONDA_LIB"/libgfortran.so.4 "$CONDA_LIB"/libquadmath.so.0; do
  ln -sf "$f" "$TMP/deps/$(basename "$f")"
done

CXXFLAGS="-O3 -march=native -std=c++17 -fopenmp -DMKL_ILP64 -m64 -w"
LIBS="-Wl,--no-as-needed $TMP/deps/libmpi.so $TMP/deps/libmkl_intel_ilp64.so $TMP/deps/libmkl_gnu_thread.so $TMP/deps/libmkl_core.so -lgomp -lpthread -lm -ldl"

# (1) unmodified reference
g++ $CXXFLAGS -I"$HERE/stubs" -I"$TMP/inc" -I"$REF" \
    "$HERE/ref_driver.cpp" "$REF/sparse_kernels.cpp" "$REF/common.cpp" "$REF/als_conjugate_gradients.cpp" $LIBS -o "$OUT/ref_driver"

# (2) patched private copy for 2.5D dense-replicate only: all headers must come from one directory
#     because the reference includes them with quotes, so mirror the headers into $TMP/patched.
for f in common.h common.cpp sparse_kernels.h sparse_kernels.cpp SpmatLocal.hpp FlexibleGrid.hpp distributed_sparse.h \
         15D_dense_shift.hpp 15D_sparse_shift.hpp 25D_cannon_dense.hpp 25D_cannon_sparse.hpp \
         als_conjugate_gradients.h als_conjugate_gradients.cpp gat.hpp json.hpp; do cp "$REF/$f" "$TMP/patched/"; done
python3 - "$TMP/patched/SpmatLocal.hpp" <<'PY'
import sys, re
p = sys.argv[1]; s = open(p).read()
old = "else if (mode == coo || mode == both) {\n\t\t\tMPI_Wait(&rRequestSend"
assert s.count(old) == 1, "patch anchor not found"
open(p, "w").write(s.replace(old, "if (mode == coo || mode == both) {\n\t\t\tMPI_Wait(&rRequestSend"))
PY
g++ $CXXFLAGS -I"$HERE/stubs" -I"$TMP/inc" -I"$TMP/patched" \
    "$HERE/ref_driver.cpp" "$TMP/patched/sparse_kernels.cpp" "$TMP/patched/common.cpp" "$TMP/patched/als_conjugate_gradients.cpp" $LIBS -o "$OUT/ref_driver_patched"

# (3) MKL ABI known-answer test (enum values / argument order of the hand-written mkl_spblas.h)
g++ -O1 -std=c++17 -DMKL_ILP64 -I"$HERE/stubs" "$HERE/mkl_kat.cpp" $LIBS -o "$OUT/mkl_kat"
echo "build_ref: ok -> $OUT"
