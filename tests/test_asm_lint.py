"""tools/asm_lint.py: the hazard lint finds what it is meant to find, and finds nothing in the shipped kernels."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("asm_lint", os.path.join(ROOT, "tools", "asm_lint.py"))
asm_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(asm_lint)


def findings(body):
    text = "kernel:\n" + body
    return [(rule, need, waited) for _, rule, need, waited, _, _ in asm_lint.lint(text.splitlines())]


def test_the_round3_incident_is_found():
    """a hand-fused v_lshl_or_b32 reading a v_dot4_u32_u8 result (a519f11)"""
    bad = """
	v_dot4_u32_u8 v5, v1, v2, v3
	;;#ASMSTART
	v_lshl_or_b32 v6, v5, 1, v7
	;;#ASMEND
"""
    assert findings(bad) == [("DOT write -> VALU read", 3, 0)]
    ok = bad.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 2\n")
    assert findings(ok) == []
    # two unrelated instructions and one state of s_nop between them: still one short... no: 2 + 1 = 3 wait states
    padded = bad.replace(";;#ASMSTART\n", "v_mov_b32_e32 v9, v8\n\tv_mov_b32_e32 v10, v8\n\ts_nop 0\n\t;;#ASMSTART\n")
    assert findings(padded) == []
    # compiler code on both ends is the compiler's business
    assert findings("\tv_dot4_u32_u8 v5, v1, v2, v3\n\tv_lshl_or_b32 v6, v5, 1, v7\n") == []


def test_every_rule_fires():
    cases = {
        "TRANS write -> VALU read": "\t;;#ASMSTART\n\tv_exp_f32_e32 v1, v2\n\t;;#ASMEND\n\tv_add_f32_e32 v3, v1, v1\n",
        "VALU writes SGPR -> VMEM reads it": "\t;;#ASMSTART\n\tv_readfirstlane_b32 s4, v1\n\tbuffer_load_dword v2, v3, s[8:11], s4 offen\n\t;;#ASMEND\n",
        "VALU writes SGPR -> lane select of v_readlane/v_writelane": "\t;;#ASMSTART\n\tv_readfirstlane_b32 s4, v1\n\tv_readlane_b32 s5, v2, s4\n\t;;#ASMEND\n",
        "VALU writes EXEC -> DPP": "\t;;#ASMSTART\n\tv_cmpx_eq_u32_e64 v1, v2\n\t;;#ASMEND\n\tv_add_u32_dpp v3, v4, v4 row_shr:1 row_mask:0xf bank_mask:0xf\n",
        "VALU writes VGPR -> DPP reads it": "\t;;#ASMSTART\n\tv_mbcnt_hi_u32_b32 v1, -1, v1\n\t;;#ASMEND\n\ts_nop 0\n\tv_add_u32_dpp v3, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n",
        "VALU writes VGPR -> v_readlane/v_readfirstlane reads it": "\tv_mov_b32_e32 v1, v2\n\t;;#ASMSTART\n\tv_readlane_b32 s5, v1, 3\n\t;;#ASMEND\n",
        "VALU writes VCC -> v_div_fmas": "\t;;#ASMSTART\n\tv_cmp_eq_u32_e32 vcc, v1, v2\n\t;;#ASMEND\n\tv_div_fmas_f32 v3, v4, v5, v6\n",
        "SALU writes M0 -> LDS-DMA / add-TID / GDS": "\t;;#ASMSTART\n\ts_mov_b32 m0, s4\n\tglobal_load_lds_dwordx4 v[2:3], off lds\n\t;;#ASMEND\n",
        "DOT write -> VALU write": "\t;;#ASMSTART\n\tv_dot4_u32_u8 v5, v1, v2, v3\n\tv_mov_b32_e32 v5, v9\n\t;;#ASMEND\n",
    }
    for rule, body in cases.items():
        got = [r for r, _, _ in findings(body)]
        assert rule in got, (rule, got)
    # a carry-out parked in an SGPR pair and read by the scalar unit (the window and append blocks): no hazard
    assert findings("\t;;#ASMSTART\n\tv_add_co_u32_e64 v1, s[4:5], v1, v1\n\ts_mov_b64 exec, s[4:5]\n\tds_add_u32 v2, v3\n\t;;#ASMEND\n") == []
    # a label ends the straight line
    assert findings("\tv_dot4_u32_u8 v5, v1, v2, v3\n.LBB0_2:\n\t;;#ASMSTART\n\tv_lshl_or_b32 v6, v5, 1, v7\n\t;;#ASMEND\n") == []


DIVERGED = """
walker:
.LBB0_1:                                ; =>This Loop Header: Depth=1
                                        ;     Child Loop BB0_2 Depth 2
	s_mov_b64 s[42:43], s[4:5]
.LBB0_2:                                ;   Parent Loop BB0_1 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
	v_mov_b32_e32 v3, 0
	s_nop 1
	v_mov_b32_dpp v3, v33 row_shr:1 row_mask:0xf bank_mask:0xf
	v_cmp_lt_u32_e64 s[44:45], s67, v33
; %bb.3:                                ;   in Loop: Header=BB0_2 Depth=2
	s_or_b64 s[36:37], s[44:45], s[36:37]
	s_andn2_b64 exec, exec, s[36:37]
	s_cbranch_execnz .LBB0_2
; %bb.4:                                ;   in Loop: Header=BB0_1 Depth=1
	s_or_b64 exec, exec, s[36:37]
	global_atomic_add v3, v22, s[18:19]
	s_cbranch_vccnz .LBB0_1
; %bb.5:
	s_endpgm
"""


def test_cross_lane_operations_in_a_loop_the_compiler_made_divergent_are_found():
    """the walker at k = 8, 9 without its latch barrier: the lanes with a sixteenth window wait at the outer
    latch while the others run the next trip's DPP moves"""
    found = asm_lint.convergence(DIVERGED.splitlines())
    assert [(fn, h, len(ops)) for fn, h, _, ops in found] == [("walker", "BB0_2", 1)]
    # the same loop left by a scalar branch: nothing to report; a divergent loop without cross-lane operations neither
    uniform = DIVERGED.replace("\ts_andn2_b64 exec, exec, s[36:37]\n\ts_cbranch_execnz .LBB0_2", "\ts_cbranch_vccnz .LBB0_2")
    assert asm_lint.convergence(uniform.splitlines()) == []
    plain = DIVERGED.replace("v_mov_b32_dpp v3, v33 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_e32 v3, v33")
    assert asm_lint.convergence(plain.splitlines()) == []
    # an operation in a loop NESTED in the divergent one counts; one in the enclosing loop does not
    outer = DIVERGED.replace("\tglobal_atomic_add v3, v22, s[18:19]", "\tv_readlane_b32 s5, v2, 3")
    assert [h for _, h, _, _ in asm_lint.convergence(outer.splitlines())] == ["BB0_2"]
    # a hand-written block that sets EXEC and puts it back: the same finding; the compiler's own s_mov_b64 exec is its business
    by_hand = plain.replace("\tv_mov_b32_e32 v3, v33\n", "\t;;#ASMSTART\n\ts_mov_b64 exec, s[8:9]\n\tds_add_u32 v2, v3\n\ts_mov_b64 exec, s[10:11]\n\t;;#ASMEND\n")
    assert [(h, len(ops)) for _, h, _, ops in asm_lint.convergence(by_hand.splitlines())] == [("BB0_2", 2)]
    assert asm_lint.convergence(by_hand.replace("\t;;#ASMSTART\n", "").replace("\t;;#ASMEND\n", "").splitlines()) == []
    inner_only = plain.replace("\ts_cbranch_vccnz .LBB0_1", "\ts_andn2_b64 exec, exec, s[30:31]\n\ts_cbranch_execnz .LBB0_1") \
        .replace("v_mov_b32_e32 v3, v33", "ds_bpermute_b32 v3, v4, v33")
    assert sorted(h for _, h, _, _ in asm_lint.convergence(inner_only.splitlines())) == ["BB0_1", "BB0_2"]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_shipped_kernels_have_no_unpadded_asm_hazard():
    path = asm_lint.device_asm()
    with open(path) as f:
        lines = f.read().splitlines()
    assert sum(1 for l in lines if ";;#ASMSTART" in l) > 1000        # the hand-written blocks are in there
    found = asm_lint.lint(lines)
    assert not found, [(fn, rule, p.text, c.text) for fn, rule, _, _, p, c in found[:5]]
    # ... and no DPP move, v_readlane or permute in a loop that hipcc wrote as one the lanes leave one by one
    assert sum(1 for l in lines if "_dpp" in l or "v_readlane" in l or "ds_bpermute" in l) > 500
    conv = asm_lint.convergence(lines)
    assert not conv, [(fn, h, ops[:2]) for fn, h, _, ops in conv[:5]]
