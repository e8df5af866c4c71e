set -e
mkdir -p gpurun_out/r4o
timeout -k 10 150 python -m pytest tests/test_gpu_golden.py -x -q -k "rung3 or rung2 or torture" > gpurun_out/r4o/small.log 2>&1 || { tail -30 gpurun_out/r4o/small.log; exit 1; }
tail -1 gpurun_out/r4o/small.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4o/tests.log 2>&1 || { tail -40 gpurun_out/r4o/tests.log; exit 1; }
tail -2 gpurun_out/r4o/tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rung in 3 4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4o/prof$rung -o run -- python3 $R/bench.py --rung $rung --steps 10 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/r4o/prof$rung.log 2>&1
python3 - $R/gpurun_out/r4o/prof$rung/run_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'init' in r['Name'] or 'seed' in r['Name'] or 'clear' in r['Name']: print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
done
