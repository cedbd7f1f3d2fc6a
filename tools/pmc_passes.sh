#!/bin/bash
# Counter passes on one kernel (run through gpurun from the repository root): each --pmc set in its own rocprofv3 run,
# never combined with a trace domain.  Usage:
#     tools/pmc_passes.sh TAG KERNEL_LIKE -- <command that launches the kernel a few times>
# Writes gpurun_out/profiles_export/${TAG}_pmc.csv (per dispatch) and ${TAG}_pmc_means.txt (mean per counter).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIKE=$2; shift 3
OUT=$R/gpurun_out/profiles_export
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LEVEL_WAVES"
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"
  "TCC_BUSY_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
)
DBS=()
i=0
for s in "${SETS[@]}"; do
  d=$R/gpurun_out/pmc_${TAG}_$i
  rm -rf "$d"
  # shellcheck disable=SC2086
  timeout 600 rocprofv3 --pmc $s -d "$d" -o p -- "$@" > "$OUT/${TAG}_pass$i.log" 2>&1 || echo "pass $i ($s) failed, see ${TAG}_pass$i.log"
  db=$(find "$d" -name "*_results.db" | head -1)
  [ -n "$db" ] && DBS+=("$db")
  i=$((i+1))
done
python "$R/tools/rocpd_summary.py" --tag "$TAG" --pmc "${DBS[@]}" --kernel-like "$LIKE" --out "$OUT" | tee "$OUT/${TAG}_pmc_means.txt"
for j in $(seq 0 $((i-1))); do rm -rf "$R/gpurun_out/pmc_${TAG}_$j"; tail -c 300 "$OUT/${TAG}_pass$j.log" | grep -i "error\|fail" ; done
true
