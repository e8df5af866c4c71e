ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r4h"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/g$i" -o run -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu --no-extras $BENCH_ARGS > "$OUT/g$i.log" 2>&1 && echo "group $i ok" || echo "group $i FAILED"
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if "FETCH_SIZE" in v: print(k[:50], "fetch MB", round(v["FETCH_SIZE"]*1024/1e6,1), "write MB", round(v.get("WRITE_SIZE",0)*1024/1e6,1), "total(2F+W)", round((2*v["FETCH_SIZE"]+v.get("WRITE_SIZE",0))*1024/1e6,1))
PY
rm -rf "$OUT"/g1 "$OUT"/g2
