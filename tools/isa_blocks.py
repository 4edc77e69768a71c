#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel from `llvm-objdump -d` output of a gfx950 code
object: VALU instructions by encoding size (4 / 8 bytes -- the issue cost on gfx950 follows the
encoding size, tools/issue_rate.hip), SALU, LDS, VMEM, waits.  Usage: isa_blocks.py file.dis"""
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)(?:<.*>)?\s*$", line)
    if m:
        op, args, addr, enc = m.group(1), m.group(2), int(m.group(3), 16), m.group(4).split()
        rows.append((addr, op, args, 4 * len(enc)))
targets = set()
for addr, op, args, sz in rows:
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"(-?\d+)", args)
        if m:
            d = int(m.group(1))
            targets.add(addr + 4 + 4 * (d - 65536 if d >= 32768 else d))
blk = None
out = []
for addr, op, args, sz in rows:
    if blk is None or addr in targets:
        blk = {"start": addr, "v4": 0, "v8": 0, "s": 0, "lds": 0, "vmem": 0, "wait": 0, "n": 0, "bytes": 0, "end": ""}
        out.append(blk)
    blk["n"] += 1
    blk["bytes"] += sz
    if op.startswith("v_"):
        blk["v8" if sz >= 8 else "v4"] += 1
    elif op.startswith("ds_"):
        blk["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        blk["vmem"] += 1
    elif op == "s_waitcnt":
        blk["wait"] += 1
        blk.setdefault("waits", []).append(args)
    else:
        blk["s"] += 1
    if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
        m = re.search(r"(-?\d+)", args)
        d = int(m.group(1)) if m else 0
        tgt = addr + 4 + 4 * (d - 65536 if d >= 32768 else d) if m else 0
        blk["end"] = f"{op} -> {tgt:X}"
        blk = None
for b in out:
    if b["n"] >= int(sys.argv[2]) if len(sys.argv) > 2 else True:
        print(f"{b['start']:X}: n={b['n']:4d} bytes={b['bytes']:5d} VALU4={b['v4']:3d} VALU8={b['v8']:3d} SALU={b['s']:3d} "
              f"LDS={b['lds']:3d} VMEM={b['vmem']:2d} waits={b.get('waits', [])} {b['end']}")
