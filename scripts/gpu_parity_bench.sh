# Bounded small test first, then the parity files, then a rung-3 and a rung-4 bench line (GPU box).
set -e
mkdir -p gpurun_out/r4o
timeout -k 10 150 python -m pytest tests/test_gpu_golden.py -x -q -k "rung3 or rung2 or torture" > gpurun_out/r4o/small.log 2>&1 || { tail -30 gpurun_out/r4o/small.log; exit 1; }
tail -2 gpurun_out/r4o/small.log
timeout -k 10 900 python -m pytest tests/test_gpu_act.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_long_horizon.py tests/test_gpu_gen.py tests/test_gpu_episode_stats.py tests/test_gpu_envs.py -x -q > gpurun_out/r4o/tests.log 2>&1 || { tail -40 gpurun_out/r4o/tests.log; exit 1; }
tail -3 gpurun_out/r4o/tests.log
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu --no-extras > gpurun_out/r4o/b3.json 2> gpurun_out/r4o/b3.err
python -c "
import json; d=json.load(open('gpurun_out/r4o/b3.json')); print('rung3', round(d['value']/1e6,1), d['ms_per_step'], d['kernels_ms'], d['variants'])"
timeout -k 10 300 python bench.py --rung 4 --steps 60 --warmup 10 --no-cpu --no-extras > gpurun_out/r4o/b4.json 2> gpurun_out/r4o/b4.err
python -c "
import json; d=json.load(open('gpurun_out/r4o/b4.json')); print('rung4', round(d['value']/1e6,1), d['ms_per_step'], d['kernels_ms'], d['variants'])"
