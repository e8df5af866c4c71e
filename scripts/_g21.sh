set -e
mkdir -p gpurun_out/r4o
# small first, bounded: a hang shows here within a minute
timeout -k 10 150 python -m pytest tests/test_gpu_golden.py -x -q -k "rung3 or rung2 or torture" > gpurun_out/r4o/small.log 2>&1 || { tail -30 gpurun_out/r4o/small.log; exit 1; }
tail -2 gpurun_out/r4o/small.log
timeout -k 10 900 python -m pytest tests/test_gpu_act.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_long_horizon.py tests/test_gpu_gen.py tests/test_gpu_episode_stats.py tests/test_gpu_envs.py tests/test_gpu_jit.py -x -q > gpurun_out/r4o/tests.log 2>&1 || { tail -40 gpurun_out/r4o/tests.log; exit 1; }
tail -3 gpurun_out/r4o/tests.log
for v in 1 ""; do
  if [ -n "$v" ]; then export MGX_NO_SHADOW=1; else unset MGX_NO_SHADOW; fi
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu --no-extras > gpurun_out/r4o/b3.json 2> gpurun_out/r4o/b3.err
  python -c "
import json; d=json.load(open('gpurun_out/r4o/b3.json')); print('NO_SHADOW=$v', round(d['value']/1e6,1), d['ms_per_step'], d['kernels_ms']['world_actions'], d['kernels_ms']['mgx_obs_kernel'], d['variants'])"
done
