set -e
mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_gpu_episode_stats.py tests/test_gpu_envs.py tests/test_gpu_golden.py tests/test_gpu_digest.py -x -q > gpurun_out/r4c/tests.log 2>&1 || { tail -40 gpurun_out/r4c/tests.log; exit 1; }
tail -3 gpurun_out/r4c/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4c/prof -o run -- python scripts/env_throughput.py 65536 600 unchecked_actions > gpurun_out/r4c/envtp.json 2> gpurun_out/r4c/envtp.err
cat gpurun_out/r4c/envtp.json
find gpurun_out/r4c/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4c/kernel_stats.csv \;
rm -rf gpurun_out/r4c/prof
python - <<'PY'
import csv
for r in csv.reader(open('gpurun_out/r4c/kernel_stats.csv')):
    if r[0]=="Name": continue
    print(f"{r[0][:60]:60s} calls={r[1]:>5} avg_us={float(r[3])/1e3:9.1f} min={float(r[5])/1e3:8.1f} max={float(r[6])/1e3:9.1f}")
PY
timeout -k 10 300 python scripts/env_throughput.py 65536 800 > gpurun_out/r4c/envtp_all.json 2> gpurun_out/r4c/envtp_all.err
cat gpurun_out/r4c/envtp_all.json
