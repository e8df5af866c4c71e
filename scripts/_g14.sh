echo "preset AnyLength:"; python scripts/env_throughput.py 65536 600 unchecked_actions 2>/dev/null
echo "jit fixed max_steps:"; MGX_OBS_GENERIC=1 MGX_ENV_SPECIALIZE=sync python scripts/env_throughput.py 65536 600 unchecked_actions 2>/dev/null
