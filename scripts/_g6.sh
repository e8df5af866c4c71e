set -e
mkdir -p gpurun_out/r4f
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err || { tail -30 gpurun_out/r4f/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4f/bench.json'))
print("head", d["value"], d["ms_per_step"], d["variants"])
print("long", d["long_run"]["value"], d["long_run"]["ms_per_step"], d["long_run"]["rounds"]["cv"])
print("generic", d["generic"]["value"], d["generic"]["variants"], d["generic"]["kernels_ms"])
j=d.get("run_time_specialisation")
if j: print("jit generic", j["generic"]["value"], j["generic"]["variants"], "spec", j["specialised"]["value"], j["specialised"]["variants"], j["code_objects_ready_after_s"], j["errors"])
print("r4", d["rung4"]["value"], d["rung4"]["variants"])
print("wrap", d["env_wrapper"])
PY
