python -m pytest tests -m gpu -x -q -k "parity or golden or act or long" 2>&1 | tail -2
for cfg in "3 0" "3 1" "4 0"; do set -- $cfg
if [ "$2" = 1 ]; then export MGX_ACT_LEAN=1; else unset MGX_ACT_LEAN; fi
python bench.py --rung $1 --no-cpu --no-extras --steps 60 --warmup 20 > gpurun_out/vb.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/vb.json')); print('rung$1 lean_act=$2', round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['kernels_ms'].items()})"
done
