"""Timeline of ONE replayed step from a rocprofv3 kernel trace of `bench.py` in HIP-graph mode: wall time of the step, union of
the kernel intervals (chip busy with at least one kernel), time with two or more kernels in flight, idle gaps, and the gap
histogram.    python tools/graph_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5   # which optimizer launch ends the step (bench.py: 3 eager warm-ups, then the replays)
print(f"{len(idx)} optimizer launches in the trace; step = launches between #{k - 1} and #{k}")
a, b = idx[k - 1], idx[k]
step = rows[a + 1:b + 1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(step[-1]["End_Timestamp"])
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = over = 0
depth, last = 0, t0
gaps = []
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += d
    last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"launches {len(step)}  wall {(t1 - t0) / 1e6:.3f} ms  sum of kernel durations {tot / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  "
      f">=2 kernels in flight {over / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
import collections
h = collections.Counter(min(g // 1000, 20) for g in gaps)
print("gap histogram (us: count):", dict(sorted(h.items())))
