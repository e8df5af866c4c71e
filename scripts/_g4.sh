set -e
mkdir -p gpurun_out/r4d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sk in 0; do
  MGX_INIT_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4d/prof$sk -o run -- python scripts/env_throughput.py 65536 300 unchecked_actions_no_episode_stats > gpurun_out/r4d/envtp$sk.json 2> gpurun_out/r4d/envtp$sk.err
  f=$(find gpurun_out/r4d/prof$sk -name "*kernel_stats.csv")
  python - $f $sk <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'init_wave' in r[0]: print("skip", sys.argv[2], "avg_us", float(r[3])/1e3, "min", float(r[5])/1e3)
PY
  rm -rf gpurun_out/r4d/prof$sk
done
