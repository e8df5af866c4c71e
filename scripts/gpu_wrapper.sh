# Agent-steps/s through MettaGridBatchedEnv at rung 3 and rung 4, and the rung-4 kernel breakdown under rocprofv3 (GPU box).
set -e
mkdir -p gpurun_out/r4w
python scripts/env_throughput.py 65536 1500 unchecked_actions > gpurun_out/r4w/plain3.json 2> gpurun_out/r4w/plain3.err
cat gpurun_out/r4w/plain3.json
MGX_ENV_RUNG=4 timeout -k 10 600 python scripts/env_throughput.py 65536 300 unchecked_actions > gpurun_out/r4w/plain4.json 2> gpurun_out/r4w/plain4.err || { tail -5 gpurun_out/r4w/plain4.err; exit 1; }
cat gpurun_out/r4w/plain4.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
MGX_ENV_RUNG=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4w/prof4 -o run -- python3 $R/scripts/env_throughput.py 65536 150 unchecked_actions > $R/gpurun_out/r4w/prof4.log 2>&1
python3 - $R/gpurun_out/r4w/prof4/run_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
