#!/usr/bin/env bash
# Builds the C++ host layer (schedules, sparse storage, transports, C ABI) -> lib/libhnh_host.so.
# It does not link the HIP library: the kernel ABI is dlopen()ed at run time (backend.hpp).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/../lib"
build() {  # $1 = output name, rest = extra flags
    out="$1"; shift
    g++ -O2 -g -std=c++17 -fopenmp -fPIC -shared -Wall -Wno-sign-compare -Wno-unused-variable "$@" \
        -I"$ROOT/include" -I"$HERE/host" \
        "$HERE/host/world.cpp" "$HERE/host/sparse_kernels.cpp" "$HERE/host/er_generator.cpp" "$HERE/host/c_api.cpp" \
        -o "$HERE/../lib/$out" -ldl -lpthread -lrt -Wl,-Bsymbolic
}
build libhnh_host.so &
# the same library plus the paced stand-ins of the overlap measurements (tools/ only; see include/hnh_measurement_aids.h)
build libhnh_host_aids.so -DHNH_MEASUREMENT_AIDS &
wait
