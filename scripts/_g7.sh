set -e
mkdir -p gpurun_out/r4g
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_envs.py tests/test_gpu_digest.py tests/test_gpu_scale.py -x -q > gpurun_out/r4g/tests.log 2>&1 || { tail -40 gpurun_out/r4g/tests.log; exit 1; }
tail -3 gpurun_out/r4g/tests.log
timeout -k 10 600 python bench.py --steps 100 --warmup 20 --no-cpu --no-extras > gpurun_out/r4g/b3.json 2> gpurun_out/r4g/b3.err || { tail -20 gpurun_out/r4g/b3.err; exit 1; }
timeout -k 10 600 python bench.py --rung 4 --steps 100 --warmup 20 --no-cpu --no-extras > gpurun_out/r4g/b4.json 2> gpurun_out/r4g/b4.err || { tail -20 gpurun_out/r4g/b4.err; exit 1; }
python - <<'PY'
import json
for f in ("b3","b4"):
    d=json.load(open(f"gpurun_out/r4g/{f}.json"))
    print(f, round(d["value"]/1e6,1), "M", round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["kernels_ms"].items()})
PY
