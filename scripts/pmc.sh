#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace statistics and PMC passes over a short bench run.  One counter group
# per rocprofv3 process (kernel-trace only, never a sys/hip trace beside --pmc); FETCH_SIZE and WRITE_SIZE in their own
# passes (together they exceed the hardware).  Every pass is bounded by `timeout` and reports as soon as it ends.
# Usage (GPU box): [BENCH_ARGS='--rung 4'] bash scripts/pmc.sh <dir-under-gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmc}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[pmc] kernel trace"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu --no-extras $BENCH_ARGS > "$OUT/trace.log" 2>&1 || { echo "[pmc] kernel trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep -o '"value": [0-9.]*' "$OUT/trace.log" | head -1
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if timeout -k 10 180 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/g$i" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu --no-extras $BENCH_ARGS > "$OUT/g$i.log" 2>&1; then
    echo "[pmc] group $i ok: $group"
  else
    echo "[pmc] group $i FAILED: $group"; grep -m3 -i "error\|fail" "$OUT/g$i.log"
  fi
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SMEM
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
GROUPS
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json" && echo "[pmc] summary written"
