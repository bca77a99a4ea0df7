#!/usr/bin/env python3
"""Guard for the hand-written walks (stream_spmm.hip, stream_attn.hip), run on the generated assembly at every build
(mllp_amd/csrc/Makefile).  The walk of such a kernel is inline asm on FIXED registers v[first:167] (LDS read destinations
that are written asynchronously, accumulators, temporaries of the DPP exchanges).  Checked here:
  1. no compiler-generated instruction names a register >= first (the kernel is compiled with amdgpu_num_vgpr(first),
     which is a soft cap: this makes it a hard one) -- except between the asm markers `; SK_SLOW_BEGIN` and
     `; SK_SLOW_END` (a compiled slow path, entered when nothing of the asm's is live; the scan is linear, so a slow-path
     block that the compiler lays out elsewhere fails the build);
  2. DPP hazard (2 wait states between a VALU write of a VGPR and a DPP read of it): for every *_dpp instruction inside
     an asm statement, neither of the two instructions in front of it (inside the asm or before it) is a VALU instruction
     writing its DPP source (src0);
  3. transcendental hazard (gfx940+: 1 wait state between v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos and a
     non-transcendental VALU instruction that reads the result), inside asm statements.
usage: python3 tools/check_asm_reads.py <file.s> <first reserved vgpr | substr=first,substr=first,...>  (exit 1 on a violation)
       with the second form the limit of a kernel is the one whose substring its (mangled) name contains; functions that
       match none are not checked for rule 1."""
import re, sys

path, spec = sys.argv[1], sys.argv[2]
by_name = None
if "=" in spec:
    by_name = [(k, int(v)) for k, v in (item.split("=") for item in spec.split(","))]
    first = None
else:
    first = int(spec)
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def regs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


bad, in_asm, n_asm_reads, n_dpp, slow = 0, False, 0, 0, False
prev = []          # the last two instructions: (text, is_valu, registers written, is_trans)
for no, ln in enumerate(open(path).read().splitlines(), 1):
    s = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
    if s.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if s.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if in_asm and "SK_SLOW_BEGIN" in ln:
        slow = True
    if in_asm and "SK_SLOW_END" in ln:
        slow = False
    if by_name is not None and s.startswith(".type") and "@function" in s:
        name = s.split()[1].split(",")[0]
        first = next((f for k, f in by_name if k in name), None)
    if not s or s.startswith(".") or s.endswith(":"):
        continue
    ops = s.split(None, 1)
    mnem, rest = ops[0], (ops[1] if len(ops) > 1 else "")
    fields = [f.strip() for f in rest.split(",")]
    is_valu = mnem.startswith("v_")
    is_trans = mnem.startswith(TRANS)
    written = regs(fields[0]) if is_valu and fields else set()
    if in_asm:
        if mnem.startswith("ds_read_b"):
            n_asm_reads += 1
        if mnem.endswith("_dpp"):
            n_dpp += 1
            src0 = regs(fields[1].split()[0]) if len(fields) > 1 else set()
            for ptxt, pvalu, pw, _ in prev[-2:]:
                if pvalu and (pw & src0):
                    print(f"{path}:{no}: `{s}` reads by DPP what `{ptxt}` wrote less than two instructions earlier")
                    bad += 1
        if is_valu and not is_trans and prev and prev[-1][3]:
            read = set().union(*[regs(f) for f in fields[1:]]) if len(fields) > 1 else set()
            if "fmac" in mnem or "_mac_" in mnem:
                read |= regs(fields[0])
            if read & prev[-1][2]:
                print(f"{path}:{no}: `{s}` reads the result of `{prev[-1][0]}` without a wait state")
                bad += 1
    elif not slow and first is not None:
        for r in regs(s):
            if r >= first:
                print(f"{path}:{no}: compiler-generated `{s}` names the reserved register v{r}")
                bad += 1
                break
    if mnem == "s_nop":
        n = int(rest.strip() or 0) + 1
        prev = (prev + [("s_nop", False, set(), False)] * n)[-2:]
    else:
        prev = (prev + [(s, is_valu, written, is_trans)])[-2:]
print(f"{path}: {n_asm_reads} asm reads, {n_dpp} asm DPP instructions, {bad} violation(s)")
sys.exit(1 if bad else 0)
