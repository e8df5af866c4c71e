#!/bin/bash
# PMC passes over a short bench run (one counter group per rocprofv3 process, kernel-trace only).
# Usage (GPU box): bash scripts/pmc.sh <outdir-under-gpurun_out>
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmc}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d "$OUT/g$i" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu > "$OUT/g$i.log" 2>&1
  echo "group $i done: $group"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_ANY
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_ANY
SQ_IFETCH SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
FETCH_SIZE WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
GROUPS
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json"
