# rocprofv3 kernel trace + PMC passes of the rung-3 bench -> gpurun_out/r04_rung3 (then scripts/collect_profiles.py r04).
bash scripts/pmc.sh r04_rung3
for d in gpurun_out/r04_rung3; do
  find $d -name "*counter_collection.csv" -size +20M -delete
  find $d -name "*.db" -delete
done
du -sh gpurun_out/r04_rung3
