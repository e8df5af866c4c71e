"""Copies what scripts/pmc.sh left under gpurun_out/<prefix>_rung{3,4}/ into the tracked profiles/<round>/ directory and
rebuilds profiles/pmc_traffic.json (the per-launch HBM traffic bench.py reports as roofline.traffic).
Usage: python scripts/collect_profiles.py r02"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_dir = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out_dir, exist_ok=True)
build = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
traffic = {
    "build": build,
    "command": "[BENCH_ARGS='--rung 4'] bash scripts/pmc.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes, "
               "bench.py --steps 6 --warmup 2 --no-cpu)",
    "units": "FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them (KiB); bytes = value * 1024; gfx950 correction FETCH_SIZE x2 "
             "(MI355X_MICROARCH.md, HBM section; calibrated there for wide streaming reads only - the scattered reads of these "
             "kernels are uncalibrated, so the figure is an upper bound on the read side)"}
for rung in (3, 4):
    src = os.path.join(ROOT, "gpurun_out", f"{rnd}_rung{rung}")
    d = json.load(open(os.path.join(src, "summary.json")))
    json.dump(d, open(os.path.join(out_dir, f"rung{rung}_pmc_summary.json"), "w"), indent=1)
    rows = list(csv.reader(open(os.path.join(src, "trace", "run_kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"rung{rung}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "mgx" in r[0] or "rocclr" in r[0]:   # our kernels + the runtime's copies; torch's one-off generators dropped
                w.writerow(r)
    # average kernel durations of the kernel-trace pass (ns), for the issue fractions below
    dur = {}
    for r in rows[1:]:
        if r and r[0].startswith("void mgx_obs_kernel<false"):   # (the initial-observation instance: not the step's kernel)
            continue
        try:
            dur[r[0].replace("void ", "").split("<")[0].split("(")[0].split("::")[-1]] = float(r[3])
        except (ValueError, IndexError):
            pass
    t = {}
    for k, v in d.items():
        if "FETCH_SIZE" not in v or k.startswith("void mgx_obs_kernel<false"):   # (the <false> variant: initial observations)
            continue
        name = k.replace("void ", "").split("<")[0].split("::")[-1]
        t[name] = {"FETCH_SIZE": v["FETCH_SIZE"], "WRITE_SIZE": v["WRITE_SIZE"]}
        t[name + "_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in v and dur.get(name) and name != "mgx_terr_kernel":   # (its launches are mostly idle: no meaningful average)
            # share of all SIMD cycles spent issuing vector-ALU instructions: wave-level VALU instructions x 4 cycles each
            # (a wave64 instruction occupies its SIMD16 for four cycles) over 1 024 SIMDs x the kernel's duration at the
            # 2.4 GHz peak engine clock (MI355X_MICROARCH.md) — a LOWER bound when the clock runs below peak
            t[name + "_valu_frac"] = v["SQ_INSTS_VALU"] * 4.0 / (1024 * dur[name] * 2.4)
    traffic[f"rung{rung}"] = t
    for line in open(os.path.join(src, "trace.log")):
        if line.startswith('{"metric"'):
            json.dump(json.loads(line), open(os.path.join(out_dir, f"rung{rung}_bench_under_rocprof.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for r in (3, 4):
    print(r, {k: round(v / 1e6) for k, v in traffic[f"rung{r}"].items() if k.endswith("launch")}, "MB per launch")
