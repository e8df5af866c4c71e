# rocprofv3 kernel trace + PMC passes of the rung-4 bench -> gpurun_out/r04_rung4 (then scripts/collect_profiles.py r04).
BENCH_ARGS='--rung 4' bash scripts/pmc.sh r04_rung4
for d in gpurun_out/r04_rung4; do
  find $d -name "*counter_collection.csv" -size +20M -delete
  find $d -name "*.db" -delete
done
du -sh gpurun_out/r04_rung4
