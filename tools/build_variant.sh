#!/bin/bash
# Development aid: builds the kernel library with extra -D flags into distributed_sddmm_amd/lib/libhnh_kernels_<name>.so
# (selected at run time with HNH_KERNEL_LIB_DEV=<path>, kernel-level tools only).  Usage: tools/build_variant.sh pipe0 -DHNH_PIPE=0
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; shift
SRC=$R/distributed_sddmm_amd/csrc/hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$R/include" -I"$SRC" "$@" \
  "$SRC/hnh_runtime.hip" "$SRC/hnh_kernels.hip" "$SRC/hnh_comm.hip" "$SRC/hnh_tuples.hip" \
  -o "$R/distributed_sddmm_amd/lib/libhnh_kernels_$NAME.so" -lrccl -Wl,-Bsymbolic
