# The whole GPU test suite in one process (GPU box): gpurun --timeout 1200 -- 'bash scripts/gpu_suite.sh'
mkdir -p gpurun_out/r4all
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4all/tests.log 2>&1
rc=$?
tail -30 gpurun_out/r4all/tests.log
exit $rc
