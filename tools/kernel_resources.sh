#!/bin/bash
# Print VGPR/SGPR/scratch/occupancy/LDS per kernel of one HIP source (hipcc -Rpass-analysis).
src=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c "$src" -o /tmp/_kr.o \
   -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur=t.split(":",1)[1].strip()[:70].ljust(72)
    elif any(t.startswith(k) for k in ("VGPRs:","AGPRs:","TotalSGPRs","ScratchSize","Occupancy","LDS Size")):
        cur+=" "+t.replace(" [bytes/lane]","").replace(" [bytes/block]","").replace(" [waves/SIMD]","")
if cur: print(cur)
'
