set -e
mkdir -p gpurun_out/r4j
timeout -k 10 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_envs.py tests/test_gpu_golden.py -x -q > gpurun_out/r4j/tests.log 2>&1 || { tail -40 gpurun_out/r4j/tests.log; exit 1; }
tail -3 gpurun_out/r4j/tests.log
timeout -k 10 300 python bench.py --rung 4 --steps 20 --warmup 5 --no-cpu --no-extras > gpurun_out/r4j/b4.json 2> gpurun_out/r4j/b4.err
python -c "
import json; d=json.load(open('gpurun_out/r4j/b4.json')); print(d['value'], d['kernels_ms'])"
