set -e
mkdir -p gpurun_out/r4m
timeout -k 10 1100 python -m pytest tests/test_gpu_jit.py -x -q -k extended > gpurun_out/r4m/tests.log 2>&1 || { tail -40 gpurun_out/r4m/tests.log; exit 1; }
tail -3 gpurun_out/r4m/tests.log
timeout -k 10 900 python scripts/jit_rung4.py 65536 40 > gpurun_out/r4m/jit4.json 2> gpurun_out/r4m/jit4.err || { tail -20 gpurun_out/r4m/jit4.err; exit 1; }
cat gpurun_out/r4m/jit4.json
