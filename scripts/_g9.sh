set -e
mkdir -p gpurun_out/r4i
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_envs.py tests/test_gpu_episode_stats.py tests/test_gpu_digest.py -x -q > gpurun_out/r4i/tests.log 2>&1 || { tail -40 gpurun_out/r4i/tests.log; exit 1; }
tail -3 gpurun_out/r4i/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4i/prof -o run -- python scripts/env_throughput.py 65536 400 unchecked_actions > gpurun_out/r4i/envtp.json 2> gpurun_out/r4i/envtp.err
cat gpurun_out/r4i/envtp.json
find gpurun_out/r4i/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4i/kernel_stats.csv \;
rm -rf gpurun_out/r4i/prof
python - <<'PY'
import csv
for r in csv.reader(open('gpurun_out/r4i/kernel_stats.csv')):
    if r[0]=="Name" or not ("mgx" in r[0]): continue
    print(f"{r[0][:60]:60s} calls={r[1]:>5} avg_us={float(r[3])/1e3:9.1f} min={float(r[5])/1e3:8.1f} max={float(r[6])/1e3:9.1f}")
PY
