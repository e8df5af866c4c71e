set -e
mkdir -p gpurun_out/r4n
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_long_horizon.py tests/test_gpu_gen.py tests/test_gpu_digest.py -x -q > gpurun_out/r4n/tests.log 2>&1 || { tail -40 gpurun_out/r4n/tests.log; exit 1; }
tail -3 gpurun_out/r4n/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu --no-extras > gpurun_out/r4n/b3_$i.json 2> gpurun_out/r4n/b3.err
python -c "
import json; d=json.load(open('gpurun_out/r4n/b3_$i.json')); print(round(d['value']/1e6,1), d['ms_per_step'], d['kernels_ms'])"
done
