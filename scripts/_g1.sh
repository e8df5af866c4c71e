set -e
mkdir -p gpurun_out/r4a
timeout -k 10 900 python -m pytest tests/test_gpu_episode_stats.py tests/test_gpu_envs.py -x -q > gpurun_out/r4a/tests.log 2>&1 || { tail -40 gpurun_out/r4a/tests.log; exit 1; }
tail -3 gpurun_out/r4a/tests.log
timeout -k 10 300 python scripts/env_throughput.py 65536 800 > gpurun_out/r4a/envtp.json 2> gpurun_out/r4a/envtp.err || { tail -20 gpurun_out/r4a/envtp.err; exit 1; }
cat gpurun_out/r4a/envtp.json
timeout -k 10 300 python bench.py --steps 50 --warmup 10 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err || { tail -20 gpurun_out/r4a/bench.err; exit 1; }
cat gpurun_out/r4a/bench.json
