"""Kernel timeline of a rocprofv3 --kernel-trace run: start/end of every launch per queue, overlap between queues.
usage: python scripts/timeline.py <dir with *_kernel_trace.csv> [first_n]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
skip = max(0, len(rows) // 2)
for s, e, k, q, st in rows[skip:skip + n]:
    print(f"{(s - t0) / 1e3:12.1f} {(e - t0) / 1e3:12.1f} {(e - s) / 1e3:9.1f} us  q={q} s={st}  {k}")
# overlap: total time during which kernels of >= 2 different queues are in flight
ev = []
for s, e, k, q, st in rows:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
active = {}
last = ev[0][0]
busy1 = busy2 = 0
for t, dlt, q in ev:
    nq = sum(1 for v in active.values() if v > 0)
    if nq >= 1: busy1 += t - last
    if nq >= 2: busy2 += t - last
    last = t
    active[q] = active.get(q, 0) + dlt
print(f"busy {busy1 / 1e6:.3f} ms, of which >= 2 queues in flight {busy2 / 1e6:.3f} ms")
