#!/usr/bin/env python3
"""Occupancy timeline of the pipelined bench region from a rocprofv3 --kernel-trace CSV:
python tools/timeline_pipelined.py <kernel_trace.csv>
For the last steps of the run: which kernels overlap, how long the device runs 0 / 1 / 2+ kernels, per-lane gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "smx::" not in n:
        continue
    short = n.split("smx::")[1].split("<")[0].split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id") or r.get("Queue_Id")))
ev.sort()
ev = ev[len(ev) * 2 // 3:]          # the last third: steady state
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"{len(ev)} kernels over {(t1 - t0) / 1e6:.3f} ms")
pts = []
for s, e, k, q in ev:
    pts.append((s, 1, k))
    pts.append((e, -1, k))
pts.sort()
running = defaultdict(int)
hist = defaultdict(int)
combo = defaultdict(int)
last = pts[0][0]
for t, d, k in pts:
    n = sum(running.values())
    hist[min(n, 4)] += t - last
    key = "+".join(sorted(kk for kk, c in running.items() for _ in range(c)))
    combo[key] += t - last
    last = t
    running[k] += d
tot = sum(hist.values())
print("kernels in flight:", {k: f"{100 * v / tot:.1f}%" for k, v in sorted(hist.items())})
for k, v in sorted(combo.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {100 * v / tot:5.1f}%  {k or '(idle)'}")
byq = defaultdict(list)
for s, e, k, q in ev:
    byq[q].append((s, e, k))
for q, lst in byq.items():
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    dur = defaultdict(list)
    for s, e, k in lst:
        dur[k].append(e - s)
    print(f"queue {q}: {len(lst)} kernels, mean gap {sum(gaps) / max(len(gaps), 1) / 1e3:.1f} us, "
          + ", ".join(f"{k} {sum(v) / len(v) / 1e3:.0f} us" for k, v in dur.items()))
