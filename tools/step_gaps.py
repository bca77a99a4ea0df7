#!/usr/bin/env python3
"""Per-launch timeline of the last training steps in a rocprofv3 --kernel-trace CSV: duration of every kernel and the
gap to the previous kernel's end, averaged over the steps (a step = the launches between two `adam_kernel`s).
usage: python3 tools/step_gaps.py <rocprof dir> [steps to average]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
n_avg = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void mllp::", "").replace("mllp::", "")
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
steps = [(ends[k] + 1, ends[k + 1] + 1) for k in range(len(ends) - 1)][-n_avg:]
L = steps[-1][1] - steps[-1][0]
steps = [s for s in steps if s[1] - s[0] == L]
acc = [[0.0, 0.0] for _ in range(L)]
span = 0.0
for a, b in steps:
    for j in range(L):
        r = rows[a + j]
        acc[j][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[j][1] += (int(r["Start_Timestamp"]) - int(rows[a + j - 1]["End_Timestamp"])) / 1e3
    span += (int(rows[b - 1]["End_Timestamp"]) - int(rows[a - 1]["End_Timestamp"])) / 1e3
n = len(steps)
print(f"{n} steps of {L} launches; step span {span / n:.1f} us")
print(f"{'kernel':40s} {'dur us':>8s} {'gap us':>8s}")
for j in range(L):
    print(f"{name(rows[steps[-1][0] + j])[:40]:40s} {acc[j][0] / n:8.1f} {acc[j][1] / n:8.1f}")
print(f"{'sum':40s} {sum(a[0] for a in acc) / n:8.1f} {sum(a[1] for a in acc) / n:8.1f}")
