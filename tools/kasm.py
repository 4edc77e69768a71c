#!/usr/bin/env python3
"""One kernel's instructions out of `hipcc -S` device assembly, with a census: tools/kasm.py file.s <substring of the mangled name> [--dump]"""
import collections
import re
import sys


def kernel(path, name):
    out, on = [], False
    for l in open(path).read().splitlines():
        if not on and re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l):
            on = True
        if on:
            out.append(l)
            if l.strip().startswith(".end_amdhsa_kernel") or l.strip().startswith(".Lfunc_end"):
                break
    return out


if __name__ == "__main__":
    k = kernel(sys.argv[1], sys.argv[2])
    ins = [l.strip() for l in k if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter()
    for l in ins:
        op = l.split()[0]
        c["scratch" if op.startswith("scratch_") else "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else
          "VMEM" if op.startswith(("buffer_", "global_", "flat_")) else "wait" if op == "s_waitcnt" else "SALU"] += 1
    print(len(ins), dict(c))
    for i, l in enumerate(ins):
        if l.startswith("scratch_"):
            print(i, l)
    if "--dump" in sys.argv:
        print("\n".join(k))
