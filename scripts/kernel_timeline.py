#!/usr/bin/env python3
"""Timeline of ONE search step from a rocprofv3 --kernel-trace CSV: for every kernel its start offset, duration and the gap
to the previous kernel's end, plus the sum of kernel time, the sum of gaps and the step's wall time. Answers "where does
the step time go that no kernel accounts for" (~0.5 ms of 6.0).
usage: kernel_timeline.py <kernel_trace.csv> [anchor kernel substring = pq_filter4_kernel] [which occurrence = -3]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "pq_filter4_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
if len(idx) < 4:
    sys.exit(f"anchor {anchor} found {len(idx)} times")
a, b = idx[which - 1], idx[which]   # from one anchor launch to the next = one step
step = ks[a:b]
t0 = step[0][0]
tot_k = tot_gap = 0
prev_end = None
print(f"step of {len(step)} kernels, wall {(ks[b][0] - t0) / 1e6:.3f} ms (anchor to anchor)")
for s, e, n in step:
    gap = 0 if prev_end is None else s - prev_end
    tot_k += e - s
    tot_gap += max(gap, 0)
    short = n.replace("cuvs_amd::(anonymous namespace)::", "").replace("cuvs_amd::", "")[:90]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short}")
    prev_end = max(e, prev_end or 0)
last_gap = ks[b][0] - prev_end
print(f"kernel time {tot_k / 1e6:.3f} ms, gaps {(tot_gap + max(last_gap, 0)) / 1e6:.3f} ms (last gap {last_gap / 1e3:.1f} us)")
