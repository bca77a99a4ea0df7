#!/usr/bin/env python3
"""Guard for stream_spmm.hip: the registers that the hand-written LDS reads land in (quads named in the asm text,
v[first:167]) must not appear in any compiler-generated instruction while such a read may be in flight, i.e. between an
asm `ds_read_b128` and the asm `s_waitcnt lgkmcnt(0)` that ends the pass (linear scan of the assembly: the sites of a
pass are laid out in program order) -- the data lands there asynchronously and only the asm's counted waits order its
use.  Elsewhere the asm statements' clobber lists keep the compiler from holding live values in them.  usage: python3 tools/check_asm_reads.py <file.s> <first reserved vgpr>  (exit 1 on a
violation)"""
import re, sys

path, first = sys.argv[1], int(sys.argv[2])
bad, in_asm, n_asm_reads, in_flight = 0, False, 0, False
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
for no, ln in enumerate(open(path).read().splitlines(), 1):
    s = ln.strip()
    if s.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if s.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    if in_asm:
        if s.startswith("ds_read_b128"):
            n_asm_reads += 1
            in_flight = True
        if s.startswith("s_waitcnt lgkmcnt(0)"):
            in_flight = False
        continue
    if not in_flight:
        continue
    for m in reg_re.finditer(s):
        hi = int(m.group(2)) if m.group(1) else int(m.group(3))
        if hi >= first:
            print(f"{path}:{no}: compiler-generated `{s}` names a reserved register")
            bad += 1
print(f"{path}: {n_asm_reads} asm reads, {bad} violation(s)")
sys.exit(1 if bad else 0)
