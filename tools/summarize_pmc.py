#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter over all passes under a directory (rocprofv3 --pmc CSV)."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "sweep_kernel"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void mllp::", "").replace("mllp::", "")
            if filt not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} mean={sum(v)/len(v):.4g}  n={len(v)}")
