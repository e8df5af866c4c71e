set -e
cd /root/repo && mkdir -p gpurun_out
MGX_WORLD_LPW=16 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity_lpw16.log 2>&1 || { tail -20 gpurun_out/parity_lpw16.log; exit 1; }
tail -2 gpurun_out/parity_lpw16.log
for cfg in "64 0" "32 0" "16 0" "8 0" "16 4" "8 4"; do
  set -- $cfg
  echo "== lpw=$1 wpe=$2" >> gpurun_out/sweep.log
  MGX_WORLD_LPW=$1 MGX_WORLD_WPE=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu >> gpurun_out/sweep.log 2>&1
done
grep -E "==|kernel_ms|\"value\"" gpurun_out/sweep.log | cut -c1-400
