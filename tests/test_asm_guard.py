"""tools/check_asm_reads.py is the build-time guard behind the hand-written walks (stream_spmm.hip, stream_attn.hip):
it must reject (1) a compiler-generated instruction that names a register the asm owns and (2) a DPP read of a VGPR that
a VALU instruction wrote less than two instructions earlier.  A GPU fault of round 3 came from exactly (1)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TOOL = os.path.join(ROOT, "tools", "check_asm_reads.py")

GOOD = """
	.text
kernel:
	v_mov_b32_e32 v3, v2
	;;#ASMSTART
	v_and_b32 v135, 0xffff, v3
	s_nop 1
	v_add_u32_dpp v134, v135, v7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1
	ds_read_b128 v[164:167], v134
	;;#ASMEND
	v_add_f32_e32 v4, v3, v2
	s_endpgm
"""
# the compiler touches v120, which the asm owns (first reserved register: 114)
BAD_RESERVED = GOOD.replace("v_add_f32_e32 v4, v3, v2", "v_add_f32_e32 v120, v3, v2")
# a VALU write of v135 one instruction in front of the DPP read of v135
BAD_DPP = GOOD.replace("\ts_nop 1\n", "\tv_mov_b32 v9, v8\n")
# the same register named inside the slow-path markers is allowed
OK_SLOW = GOOD.replace("v_add_f32_e32 v4, v3, v2",
                       ";;#ASMSTART\n\t; SK_SLOW_BEGIN\n\t;;#ASMEND\n\tv_add_f32_e32 v120, v3, v2\n"
                       "\t;;#ASMSTART\n\t; SK_SLOW_END\n\t;;#ASMEND")


def run(tmp_path, text, first=114):
    p = tmp_path / "k.s"
    p.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(p), str(first)], capture_output=True, text=True)


def test_guard_accepts_clean_assembly(tmp_path):
    r = run(tmp_path, GOOD)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 asm reads, 1 asm DPP instructions, 0 violation" in r.stdout


def test_guard_rejects_reserved_register_outside_asm(tmp_path):
    r = run(tmp_path, BAD_RESERVED)
    assert r.returncode == 1 and "reserved register v120" in r.stdout


def test_guard_rejects_dpp_read_behind_valu_write(tmp_path):
    r = run(tmp_path, BAD_DPP)
    assert r.returncode == 1 and "reads by DPP" in r.stdout
    # with two instructions in between it passes
    ok = BAD_DPP.replace("\tv_mov_b32 v9, v8\n", "\tv_mov_b32 v9, v8\n\tv_mov_b32 v10, v8\n")
    assert run(tmp_path, ok).returncode == 0


def test_guard_allows_slow_path_block(tmp_path):
    assert run(tmp_path, OK_SLOW).returncode == 0
