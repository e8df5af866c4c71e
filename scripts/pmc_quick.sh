#!/bin/bash
# Two PMC passes (instruction mix, wait / active cycles) over a short bench run, for one kernel under development.
# Usage (GPU box): [BENCH_ARGS='--rung 4'] bash scripts/pmc_quick.sh <dir-under-gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmcq}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if timeout -k 10 180 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/g$i" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 30 --no-cpu --no-extras $BENCH_ARGS > "$OUT/g$i.log" 2>&1; then
    echo "[pmc] group $i ok: $group"
  else
    echo "[pmc] group $i FAILED: $group"; grep -m3 -i "error\|fail" "$OUT/g$i.log"
  fi
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SMEM
GROUPS
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json" && echo "[pmc] summary written"
