#!/bin/bash
# Memory-pipeline counters (address unit, vector L1, its TLB) over a short bench run, one pass per group.
# Usage (GPU box): [BENCH_ARGS='--rung 4'] bash scripts/pmc_mem.sh <dir-under-gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmcm}"
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
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
GROUPS
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json" && echo "[pmc] summary written"
