set -e
mkdir -p gpurun_out/r4k
timeout -k 10 900 python -m pytest tests/test_infos_fixture.py tests/test_gpu_episode_stats.py -x -q -m gpu > gpurun_out/r4k/tests.log 2>&1 || { tail -40 gpurun_out/r4k/tests.log; exit 1; }
tail -3 gpurun_out/r4k/tests.log
