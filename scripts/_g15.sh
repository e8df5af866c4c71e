set -e
mkdir -p gpurun_out/r4l
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_envs.py tests/test_gpu_scale.py tests/test_gpu_long_horizon.py tests/test_gpu_reference_config.py -x -q > gpurun_out/r4l/tests.log 2>&1 || { tail -40 gpurun_out/r4l/tests.log; exit 1; }
tail -3 gpurun_out/r4l/tests.log
timeout -k 10 300 python bench.py --rung 4 --steps 100 --warmup 20 --no-cpu --no-extras > gpurun_out/r4l/b4.json 2> gpurun_out/r4l/b4.err
python -c "
import json; d=json.load(open('gpurun_out/r4l/b4.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
