set -e
mkdir -p gpurun_out/r4e
timeout -k 10 1100 python -m pytest tests/test_gpu_jit.py -x -q > gpurun_out/r4e/tests.log 2>&1 || { tail -60 gpurun_out/r4e/tests.log; exit 1; }
tail -3 gpurun_out/r4e/tests.log
