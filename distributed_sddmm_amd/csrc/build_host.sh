#!/usr/bin/env bash
# Builds the C++ host layer (schedules, sparse storage, transports, C ABI) -> lib/libhnh_host.so.
# It does not link the HIP library: the kernel ABI is dlopen()ed at run time (backend.hpp).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/../lib"
g++ -O2 -g -std=c++17 -fopenmp -fPIC -shared -Wall -Wno-sign-compare -Wno-unused-variable \
    -I"$ROOT/include" -I"$HERE/host" \
    "$HERE/host/world.cpp" "$HERE/host/sparse_kernels.cpp" "$HERE/host/er_generator.cpp" "$HERE/host/c_api.cpp" \
    -o "$HERE/../lib/libhnh_host.so" -ldl -lpthread -Wl,-Bsymbolic
