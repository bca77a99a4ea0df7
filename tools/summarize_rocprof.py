#!/usr/bin/env python3
"""Summarise rocprofv3 --kernel-trace CSV output: per-kernel count / avg / total, sorted by total.
usage: python3 tools/summarize_rocprof.py <dir> [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
stat = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("kernel_name")
            t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            name = name.replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*$", "", name)
            name = name.replace("void mllp::", "").replace("mllp::", "")
            s = stat[name]
            s[0] += 1; s[1] += t; s[2] = min(s[2], t); s[3] = max(s[3], t)
tot = sum(s[1] for s in stat.values()) or 1.0
lines = ["| kernel | calls | avg us | min us | max us | total us | % |", "|---|---|---|---|---|---|---|"]
for name, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| {name[:90]} | {s[0]} | {s[1]/s[0]:.2f} | {s[2]:.2f} | {s[3]:.2f} | {s[1]:.1f} | {100*s[1]/tot:.1f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        fh.write(txt + "\n")
