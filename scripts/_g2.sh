set -e
mkdir -p gpurun_out/r4b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4b/prof -o run -- python scripts/env_throughput.py 65536 600 unchecked_actions > gpurun_out/r4b/envtp.json 2> gpurun_out/r4b/envtp.err
cat gpurun_out/r4b/envtp.json
find gpurun_out/r4b/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4b/kernel_stats.csv \;
rm -rf gpurun_out/r4b/prof
cut -c1-150 gpurun_out/r4b/kernel_stats.csv | head -20
