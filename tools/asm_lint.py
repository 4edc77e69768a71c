#!/usr/bin/env python3
"""Hazard lint for the hand-written asm statements of the extension (gfx950).

hipcc's hazard recogniser pads the instruction pairs it schedules itself; it does not look inside an
inline-asm statement, and after one it only adds its fixed boundary pad.  A pair whose producer OR
consumer sits inside `;;#ASMSTART` .. `;;#ASMEND` therefore needs its wait states written by hand
(round 3's incident: a hand-fused `v_lshl_or_b32` read `v_dot4_u32_u8` results three wait states too
early -- stale masks, caught only because a later test refused the garbage).  This script walks the
device assembly (`hipcc -S --cuda-device-only`) and reports every such pair that has fewer wait states
between producer and consumer than the gfx90a/gfx940 rules ask for:

  DOT      v_dot*  writes a VGPR  ->  another VALU reads it (3) or writes it (4)
  TRANS    v_exp/log/rcp/rsq/sqrt/sin/cos writes a VGPR  ->  a non-trans VALU reads it (1)
  SGPR-VM  a VALU writes an SGPR (v_cmp*_e64, carry-out, v_readlane, v_readfirstlane)  ->  a buffer_/global_/
           flat_/scratch_ instruction reads it (5)
  SGPR-LN  a VALU writes an SGPR / VCC  ->  v_readlane / v_writelane takes it as the lane select (4)
  EXEC-DPP a VALU writes EXEC (v_cmpx)  ->  a DPP instruction (5)
  VGPR-DPP a VALU writes a VGPR  ->  a DPP instruction reads it (2)
  VGPR-RL  a VALU writes a VGPR  ->  v_readlane / v_readfirstlane reads it (1)
  M0-LDS   an SALU writes M0  ->  an LDS-DMA / add-TID / GDS instruction (1)
  VCC-FMAS a VALU writes VCC  ->  v_div_fmas (4)

Wait states: every instruction between the two counts one, `s_nop N` counts N + 1.  Pairs whose two
ends are both compiler code are the compiler's business and are not reported.  Branch targets reset
the window (a label: the straight-line distance no longer holds).

usage: tools/asm_lint.py [file.s]      (no argument: compiles varkoder_amd/csrc/vkimg.hip first)
exit status 1 when something is found.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)(_|$)")
VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_|tbuffer_)")
DPP = re.compile(r"\b(quad_perm:|row_shl:|row_shr:|row_ror:|wave_shl:|wave_shr:|wave_rol:|wave_ror:|row_mirror|row_half_mirror|row_bcast:|row_newbcast:)")
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0)(_lo|_hi)?\b")


def regs(text):
    """set of register names in an operand string: v3, s[4:5] -> {s4, s5}, vcc, exec, m0"""
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add(m.group(3) + str(i))
        else:
            out.add(m.group(6))
    return out


class Ins:
    __slots__ = ("op", "defs", "uses", "text", "line", "in_asm", "nops", "is_valu", "is_dpp", "lanesel")


def parse(op, rest, text):
    ins = Ins()
    ins.op, ins.text = op, text
    ins.nops = 0
    ins.lanesel = set()
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    # modifiers ride on the last operand ("v3 row_shr:1 row_mask:0xf"): regs() skips what is not a register
    ins.is_valu = op.startswith("v_") and not op.startswith("v_nop")
    ins.is_dpp = bool(DPP.search(text))
    ndst = 1
    if op == "s_nop":
        ins.nops = int(ops[0], 0) + 1 if ops else 1
        ins.defs, ins.uses = set(), set()
        return ins
    if re.match(r"^v_(add|sub|subrev)c?_co_|^v_(addc|subb|subbrev)_co_|^v_div_scale|^v_mad_(u|i)64_", op):
        ndst = 2 if not op.endswith("_e32") and not op.endswith("_dpp") and not op.endswith("_sdwa") else 1
    if op.startswith(("s_cmp", "s_cbranch", "s_waitcnt", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_branch", "s_bitcmp")) or \
            op.startswith(("ds_write", "ds_add_u", "ds_or_b", "ds_and_b", "ds_max_u", "ds_min_u")) and "rtn" not in op or \
            re.match(r"^(buffer|global|flat|scratch)_(store|atomic)", op) and "glc" not in text and " sc0" not in text:
        ndst = 0
    if op.startswith("v_cmpx"):
        ndst = 0
    ins.defs = set()
    for o in ops[:ndst]:
        ins.defs |= regs(o)
    ins.uses = set()
    for o in ops[ndst:]:
        ins.uses |= regs(o)
    if op.startswith("v_cmp") and op.endswith("_e32"):
        ins.defs = {"vcc"}
        ins.uses = set()
        for o in ops:
            ins.uses |= regs(o)
        ins.uses.discard("vcc")
    if op.startswith("v_cmpx"):
        ins.defs = {"exec"}
    if op.endswith("_e32") and re.match(r"^v_(add|sub|subrev)_co_|^v_(addc|subb|subbrev)_co_", op):
        ins.defs |= {"vcc"}
    if op.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
        ins.lanesel = regs(ops[2])
    if op.startswith("v_div_fmas"):
        ins.uses |= {"vcc"}
    return ins


def instructions(lines):
    """yield (function name, [Ins]) for every function of the assembly"""
    fn, cur, in_asm = None, [], False
    for n, raw in enumerate(lines, 1):
        line = raw.split(";;#")[0] if ";;#ASM" not in raw else raw
        s = line.strip()
        if ";;#ASMSTART" in s:
            in_asm = True
            continue
        if ";;#ASMEND" in s:
            in_asm = False
            continue
        s = s.split(";")[0].strip() if not s.startswith(";") else ""
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m:
            name = m.group(1)
            if not name.startswith((".L", "L")) and not re.match(r"^\d", name):
                if fn is not None:
                    yield fn, cur
                fn, cur = name, []
            else:
                cur.append(None)      # a label: the window ends here
            continue
        if re.match(r"^\d+:", s):      # local label inside an asm statement
            cur.append(None)
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        ins = parse(parts[0], parts[1] if len(parts) > 1 else "", s)
        ins.line, ins.in_asm = n, in_asm
        cur.append(ins)
    if fn is not None:
        yield fn, cur


def lint(lines):
    found = []
    for fn, seq in instructions(lines):
        for i, c in enumerate(seq):
            if c is None or c.nops:
                continue
            waited = 0
            j = i - 1
            while j >= 0 and waited < 5:
                p = seq[j]
                if p is None:
                    break
                if p.nops:
                    waited += p.nops
                    j -= 1
                    continue
                if p.in_asm or c.in_asm:
                    for rule, need in check(p, c):
                        if waited < need:
                            found.append((fn, rule, need, waited, p, c))
                waited += 1
                j -= 1
    return found


def check(p, c):
    """hazard rules between producer p and a later consumer c: (name, wait states needed)"""
    out = []
    vdefs = {r for r in p.defs if r[0] == "v" and r != "vcc"}
    sdefs = {r for r in p.defs if r[0] == "s" or r == "vcc"}
    if p.op.startswith("v_dot") and c.is_valu and not (c.op == p.op):
        if vdefs & c.uses:
            out.append(("DOT write -> VALU read", 3))
        if vdefs & c.defs:
            out.append(("DOT write -> VALU write", 4))
    if TRANS.match(p.op) and c.is_valu and not TRANS.match(c.op) and vdefs & c.uses:
        out.append(("TRANS write -> VALU read", 1))
    if p.is_valu and sdefs:
        if VMEM.match(c.op) and sdefs & c.uses:
            out.append(("VALU writes SGPR -> VMEM reads it", 5))
        if c.lanesel & sdefs:
            out.append(("VALU writes SGPR -> lane select of v_readlane/v_writelane", 4))
        if c.op.startswith("v_div_fmas") and "vcc" in sdefs:
            out.append(("VALU writes VCC -> v_div_fmas", 4))
    if p.is_valu and "exec" in p.defs and c.is_dpp:
        out.append(("VALU writes EXEC -> DPP", 5))
    if p.is_valu and c.is_dpp and vdefs & c.uses:
        out.append(("VALU writes VGPR -> DPP reads it", 2))
    if p.is_valu and c.op.startswith(("v_readlane", "v_readfirstlane")) and vdefs & c.uses:
        out.append(("VALU writes VGPR -> v_readlane/v_readfirstlane reads it", 1))
    if p.op.startswith("s_") and "m0" in p.defs and (c.op.startswith(("ds_gws", "ds_add_tid", "ds_read_addtid", "ds_write_addtid", "s_sendmsg")) or
                                                      (VMEM.match(c.op) and " lds" in c.text)):
        out.append(("SALU writes M0 -> LDS-DMA / add-TID / GDS", 1))
    return out


CROSS = re.compile(r"^(v_readlane|v_writelane|ds_bpermute|ds_permute|ds_swizzle|v_permlane)")
BLOCK = re.compile(r"^(\.LBB\d+_\d+):|^; %bb\.\d+:")
HDR = re.compile(r"Loop Header: Depth=(\d+)")
PARENT = re.compile(r"Parent Loop (BB\d+_\d+) Depth=(\d+)")
INLOOP = re.compile(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)")


def convergence(lines):
    """Cross-lane operations inside a loop that hipcc has made DIVERGENT.

    A loop whose lanes leave one by one ends in `s_andn2_b64 exec, exec, <left>` + `s_cbranch_exec(n)z` (SI_LOOP); a loop
    that is uniform in the source -- its condition a ballot, its body full of DPP moves, v_readlane, ballots -- must not
    come out that way (nor may one that holds a hand-written `s_mov_b64 exec, ...`: the window blocks set EXEC per window and put
    back what they read at their head).  It did once (the walker at k = 8, 9): the body ended in a per-lane `if`, the
    arithmetic behind the `if` was duplicated into both of its exits, the two exits became two back edges, the two back
    edges an inner and an outer loop, and the lanes that took the inner one ran the next trip's ballot and DPP moves
    while the others waited at the outer latch.  The loop structure is read from the comments hipcc writes at every block
    ("in Loop: Header=BBx_y Depth=d", "Parent Loop ...", "Loop Header: Depth=d"); instructions inside asm statements
    count as cross-lane when they are DPP, v_readlane/v_writelane, permutes, or write EXEC.  Returns [(function, header label,
    line of the loop's exec update, [cross-lane instruction texts ...])]."""
    found = []
    fn = None
    parent = {}        # header label -> enclosing header label (None at depth 1)
    divergent = {}     # header label -> line number of its exec update
    body = {}          # header label (innermost) -> [(line, text)] cross-lane instructions
    cur = None         # innermost loop of the current block
    pending = None     # a block label whose comment lines are still being read
    in_asm = False     # between ;;#ASMSTART and ;;#ASMEND

    def flush():
        for h, at in sorted(divergent.items(), key=lambda kv: kv[1]):
            inside = []
            for b, ins in body.items():
                a = b
                while a is not None and a != h:
                    a = parent.get(a)
                if a == h:
                    inside += ins
            if inside:
                found.append((fn, h, at, [t for _, t in sorted(inside)]))

    for n, raw in enumerate(lines, 1):
        s = raw.strip()
        m = re.match(r"^([A-Za-z_$][\w.$]*):", s)
        if m and not m.group(1).startswith(("L", ".L")):
            if fn is not None:
                flush()
            fn, parent, divergent, body, cur, pending, in_asm = m.group(1), {}, {}, {}, None, None, False
            continue
        b = BLOCK.match(s)
        if b or (pending and s.startswith(";") and not s.startswith(";;#")):
            if b:
                pending = (b.group(1) or "").lstrip(".L") and b.group(1)[2:] if b.group(1) else "bb"
                cur = None
                parents = []
            il = INLOOP.search(s)
            if il:
                cur = il.group(1)
            for pm in PARENT.finditer(s):
                parents.append((int(pm.group(2)), pm.group(1)))
            if HDR.search(s) and pending not in (None, "bb"):
                d = int(HDR.search(s).group(1))
                above = [lab for dep, lab in parents if dep == d - 1]
                parent[pending] = above[0] if above else None
                cur = pending
            if b and not s.startswith(";"):
                rest = s.split(":", 1)[1]
                if not rest.strip().startswith(";"):
                    pending = None
            continue
        pending = None
        if not s or s.startswith((".", ";")) and ";;#" not in s:
            continue
        if ";;#" in s:
            in_asm = ";;#ASMSTART" in s
            continue
        text = s.split(";")[0].strip()
        if not text or cur is None:
            continue
        op = text.split(None, 1)[0]
        if op == "s_andn2_b64" and re.match(r"^s_andn2_b64\s+exec,\s*exec,", text):
            divergent.setdefault(cur, n)
        elif CROSS.match(op) or DPP.search(text) or (in_asm and re.match(r"^s_mov_b64\s+exec,", text)):
            # (a hand-written block that sets EXEC per window and puts back what it read at its head assumes that every lane
            # that entered the loop is still in it)
            body.setdefault(cur, []).append((n, text))
    if fn is not None:
        flush()
    return found


def device_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="vk_lint_"), "vkimg.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-S",
           "--cuda-device-only", os.path.join(ROOT, "varkoder_amd", "csrc", "vkimg.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, cwd="/tmp")
    return out


def main(argv):
    path = argv[1] if len(argv) > 1 else device_asm()
    with open(path) as f:
        lines = f.read().splitlines()
    found = lint(lines)
    nasm = sum(1 for l in lines if ";;#ASMSTART" in l)
    for fn, rule, need, waited, p, c in found:
        print(f"{fn}: {rule}: {need} wait states needed, {waited} present")
        print(f"    line {p.line}{' (asm)' if p.in_asm else ''}: {p.text}")
        print(f"    line {c.line}{' (asm)' if c.in_asm else ''}: {c.text}")
    conv = convergence(lines)
    for fn, header, at, ops in conv:
        print(f"{fn}: loop {header} leaves its lanes one by one (line {at}) and holds {len(ops)} cross-lane operations, e.g.")
        for t in ops[:3]:
            print(f"    {t}")
    print(f"asm_lint: {nasm} asm statements, {len(found)} hazard findings, {len(conv)} divergent loops with cross-lane operations in {path}")
    return 1 if found or conv else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
