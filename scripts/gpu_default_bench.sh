# The default bench.py run end to end, timed, with the fields of its line printed (GPU box).
mkdir -p gpurun_out/r4f
( time python bench.py > gpurun_out/r4f/bench_default.json 2> gpurun_out/r4f/bench_default.err ) 2> gpurun_out/r4f/time.txt
tail -3 gpurun_out/r4f/time.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench_default.json').read().strip().splitlines()[-1])
for k in ('metric','value','unit','ms_per_step','n_gpus','steps','vs_baseline','dtype','scaling','variants','kernels_ms','roofline','cpu_baseline','long_run','generic','run_time_specialisation','env_wrapper'):
    print(k, d.get(k))
print([k for k in d.keys()])
PY
