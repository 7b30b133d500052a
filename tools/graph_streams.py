"""Per-stream view of ONE replayed step from a rocprofv3 kernel trace of `bench.py` (HIP-graph mode): busy time of each queue,
time with only the side queue running, and the longest stretches in which the main queue is idle (with what ran meanwhile).
    python tools/graph_streams.py <kernel_trace.csv> [optimizer launch index that ends the step]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
a, b = idx[k - 1], idx[k]
step = rows[a + 1:b + 1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"step = {len(step)} launches, wall {(t1 - t0) / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
main_q = max(byq, key=lambda q: len(byq[q]))
for q, rs in byq.items():
    d = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}{' (main)' if q == main_q else ''}: {len(rs)} launches, sum of durations {d / 1e6:.3f} ms, "
          f"first start +{(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.0f} us, last end +{(int(rs[-1]['End_Timestamp']) - t0) / 1e3:.0f} us")
# main-queue idle stretches
m = byq[main_q]
idle = []
last = t0
for r in m:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > last:
        idle.append((s - last, last, s))
    last = max(last, e)
if t1 > last:
    idle.append((t1 - last, last, t1))
tot_idle = sum(i[0] for i in idle)
print(f"main queue idle: {tot_idle / 1e6:.3f} ms in {len(idle)} stretches; gaps < 3 us: {sum(1 for i in idle if i[0] < 3000)}, "
      f"3-6 us: {sum(1 for i in idle if 3000 <= i[0] < 6000)}, >= 6 us: {sum(1 for i in idle if i[0] >= 6000)}")
others = [r for q, rs in byq.items() if q != main_q for r in rs]
for d, s, e in sorted(idle, reverse=True)[:12]:
    mean = [r["Kernel_Name"].split("(")[0][-40:] for r in others if int(r["Start_Timestamp"]) < e and int(r["End_Timestamp"]) > s]
    prev = [r for r in m if int(r["End_Timestamp"]) <= s]
    nxt = [r for r in m if int(r["Start_Timestamp"]) >= e]
    print(f"  {d / 1e3:7.1f} us at +{(s - t0) / 1e3:7.0f} us  after {prev[-1]['Kernel_Name'][:36] if prev else '-':36s} before {nxt[0]['Kernel_Name'][:36] if nxt else '-':36s} meanwhile: {', '.join(mean[:4])}")
# side-only time
ev = []
for r in step:
    tag = 0 if r["Queue_Id"] == main_q else 1
    ev.append((int(r["Start_Timestamp"]), 1, tag))
    ev.append((int(r["End_Timestamp"]), -1, tag))
ev.sort()
depth = [0, 0]
last = t0
both = only_main = only_side = none = 0
for t, d, tag in ev:
    dt = t - last
    if depth[0] and depth[1]: both += dt
    elif depth[0]: only_main += dt
    elif depth[1]: only_side += dt
    else: none += dt
    depth[tag] += d
    last = t
print(f"main only {only_main / 1e6:.3f} ms, both {both / 1e6:.3f} ms, side only {only_side / 1e6:.3f} ms, nothing {none / 1e6:.3f} ms")
