#!/bin/bash
# Development aid: builds the host layer + a multi-rank harness with AddressSanitizer / UBSan and runs every schedule
# (fusedSpMM both ways, sddmmA/B, spmmA/B, the GAT forward pass twice where R is not split, two ALS half-steps) on p logical ranks
# over the oracle's CPU test double — first with the defaults, then under every switch that selects another host code path.
# (ThreadSanitizer is not part of this: gcc 11's libtsan does not intercept pthread_cond_clockwait, so every
# condition_variable::wait_for of the loopback transport is reported as a double lock + races on the data it guards.)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
H=$ROOT/distributed_sddmm_amd/csrc/host
OUT=${TMPDIR:-/tmp}/hnh_sanitize
mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-sign-compare -I"$ROOT/include" -I"$H" \
    "$H/world.cpp" "$H/sparse_kernels.cpp" "$H/er_generator.cpp" "$ROOT/tools/sanitize/spmd_harness.cpp" -o "$OUT/spmd_asan" -ldl -lpthread
# the test double itself under the same sanitizers, with its stream-order checker on (oracle/hnh_stream_order.h: the checker's own code is checked too;
# a run that saw a race of the stream protocol exits with status 86 and the report)
gcc -O1 -g -std=c11 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -I"$ROOT/include" "$ROOT/oracle/hnh_oracle_backend.c" -o "$OUT/liboracle_asan.so"
for cfg in "1 1 15d_fusion2" "4 1 15d_fusion2" "4 2 15d_fusion2" "8 2 15d_fusion1" "4 1 15d_sparse" "8 2 25d_dense_replicate" "8 2 25d_sparse_replicate"; do
    ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=1 "$OUT/spmd_asan" "$ROOT/oracle/liboracle_backend.so" $cfg | tail -1
    HNH_ORDER_CHECK=1 ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=1 "$OUT/spmd_asan" "$OUT/liboracle_asan.so" $cfg | tail -1
done
for envs in HNH_RING_MODE=relay HNH_ACC_HALVES=0 HNH_SHIP_INDICES=1 HNH_BORROW=off HNH_BORROW=force HNH_MESH_CHUNKS=4 HNH_MESH_TAPER=3,4,4,3,2,1,1 \
            HNH_HOST_SETUP=1 HNH_GAT_SERIAL=1 HNH_WINDOW_MERGE=0 HNH_WINDOW_MERGE_CAP=2 HNH_ORACLE_EVENTS_PENDING=2 HNH_PERF_COUNTERS=0 \
            HNH_FUSION1_MESH=0 HNH_MESH_TAPER=1,2,2,2,1,1 HNH_MESH_CHUNKS=1 HNH_RMAT=0.57,0.19,0.19; do
    for cfg in "4 1 15d_fusion2" "4 2 15d_fusion1" "8 2 25d_dense_replicate" "4 1 15d_sparse"; do
        echo -n "$envs: "
        env "$envs" ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=1 "$OUT/spmd_asan" "$ROOT/oracle/liboracle_backend.so" $cfg | tail -1
    done
done

# Second stage: the ipc-pull transport's host side (IpcWorld: shared-memory control plane, mailboxes, flag words, handle caches) across
# PROCESSES — examples/verify built with the same sanitizers, as 4 / 8 ranks over the test double's process_vm_readv pull.
g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-sign-compare -I"$ROOT/include" -I"$H" \
    "$H/world.cpp" "$H/sparse_kernels.cpp" "$H/er_generator.cpp" "$ROOT/examples/verify.cpp" -o "$OUT/verify_asan" -ldl -lpthread -lrt
cp "$ROOT/oracle/liboracle_backend.so" "$OUT/libhnh_kernels.so"   # (the drivers load the kernel library next to the host code: here, the binary)
for cfg in "4 2 15d_fusion1" "4 1 15d_fusion2" "4 1 15d_sparse" "4 1 25d_dense_replicate" "8 2 25d_sparse_replicate"; do
    set -- $cfg
    pids=()
    for r in $(seq 0 $(($1 - 1))); do
        RANK=$r WORLD_SIZE=$1 LOCAL_RANK=$r HNH_DEVICE=0 HNH_TRANSPORT=ipc HNH_IPC_SESSION="asan_$$_$3" HNH_IPC_WAIT_S=120 HNH_HOST_SETUP=1 \
            ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=1 "$OUT/verify_asan" er:8:6 "$3" 16 "$2" > "$OUT/ipc_rank$r.log" 2>&1 &
        pids+=($!)
    done
    for p in "${pids[@]}"; do wait "$p"; done
    echo "ipc $cfg: $(grep -h Fingerprint "$OUT/ipc_rank0.log" | tr '\n' ' ')"
done

# Third stage: RcclWorld — the default transport (explicit-peer groups, native collectives on the world, host data staged through device
# memory) — the same way, over the test double's emulation of RCCL between processes; the unique id travels through the launch-named file.
cp "$OUT/liboracle_asan.so" "$OUT/libhnh_kernels.so"   # (sanitized double: its emulation and stream-order checker are under the sanitizers too)
for cfg in "4 2 15d_fusion1" "4 1 15d_fusion2" "4 1 15d_sparse" "4 1 25d_dense_replicate" "8 2 25d_sparse_replicate"; do
    set -- $cfg
    pids=()
    for r in $(seq 0 $(($1 - 1))); do
        RANK=$r WORLD_SIZE=$1 LOCAL_RANK=$r HNH_DEVICE=0 HNH_JOB_TOKEN="asan_$$_$3" HNH_ORACLE_COMM_WAIT_S=120 HNH_HOST_SETUP=1 HNH_ORDER_CHECK=1 \
            ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=1 "$OUT/verify_asan" er:8:6 "$3" 16 "$2" > "$OUT/rccl_rank$r.log" 2>&1 &
        pids+=($!)
    done
    for p in "${pids[@]}"; do wait "$p"; done
    echo "rccl $cfg: $(grep -h Fingerprint "$OUT/rccl_rank0.log" | tr '\n' ' ')"
done
