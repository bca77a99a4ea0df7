#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: per training step of tools/profile_step.py netlib, how much of the wall time has
0 / 1 / >= 2 kernels running, and the busiest kernels.  usage: python3 tools/timeline.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void mllp::", "").replace("mllp::", "")))
rows.sort()
# steps end with adam_kernel; take the last 5 complete steps
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
lo, hi = adam[-6] + 1, adam[-1] + 1
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
ev = []
for s, e, _ in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cov = defaultdict(int); cur = 0; last = t0
for t, d in ev:
    cov[min(cur, 2)] += t - last; last = t; cur += d
tot = t1 - t0
print(f"5 steps: {tot / 5e3:.1f} us per step; kernels running: none {100 * cov[0] / tot:.1f} %, one {100 * cov[1] / tot:.1f} %, two or more {100 * cov[2] / tot:.1f} %")
dur = defaultdict(float)
for s, e, n in seg:
    dur[n] += (e - s) / 5e3
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {d:7.1f} us/step  {n[:70]}")
print(f"   {sum(dur.values()):7.1f} us/step  sum of kernel durations")
