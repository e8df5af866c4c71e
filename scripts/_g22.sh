set -e
mkdir -p gpurun_out/r4w
R=$GRAFT_REPO_ROOT
python scripts/env_throughput.py 65536 1500 unchecked_actions > gpurun_out/r4w/plain.json 2> gpurun_out/r4w/plain.err
cat gpurun_out/r4w/plain.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4w/prof -- python3 $R/scripts/env_throughput.py 65536 600 unchecked_actions > $R/gpurun_out/r4w/prof.log 2>&1
cd $R
f=$(find gpurun_out/r4w/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
