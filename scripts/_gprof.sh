bash scripts/pmc.sh r04_rung3 && BENCH_ARGS='--rung 4' bash scripts/pmc.sh r04_rung4
ls gpurun_out/r04_rung3 gpurun_out/r04_rung4 | head -40
# keep only what collect_profiles.py needs (the counter CSVs are large)
for d in gpurun_out/r04_rung3 gpurun_out/r04_rung4; do
  find $d -name "*counter_collection.csv" -delete
  find $d -name "*kernel_trace.csv" -delete
  find $d -name "*.db" -delete
done
du -sh gpurun_out/r04_rung3 gpurun_out/r04_rung4
