#!/usr/bin/env python3
"""Experiment: put an s_nop 0 between vector instructions of different issue classes in a compiler-made listing.
    python3 tools/experiments/asm_idle_slots.py in.s out.s <mode> kernel-name-substring ...
mode: sf   = after a 4-cycle-class instruction that is directly followed by a 2-cycle-class one
      both = at every change of class
      pair = after every second vector instruction of a run that mixes the classes
Only the named kernels are touched.  The class table is the one of tools/isa_phase_table.py."""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import isa_phase_table as T


def cls(line):
    t = line.strip()
    if not t.startswith("v_"):
        return None
    m = re.sub(r"_(e32|e64|dpp|sdwa)$", "", t.split()[0])
    if m.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane")):
        return None                                  # (results go to scalar registers / need their own hazards: leave alone)
    return "F" if T.valu_cost(m) < 1.1 else "S"


def main(src, dst, mode, names):
    out, inside, prev, added = [], False, None, 0
    run = 0
    for line in open(src):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = any(n in m.group(1) for n in names)
            prev = None
        if line.startswith(".Lfunc_end"):
            inside = False
        if inside and line.startswith("\t") and not line.strip().startswith((".", ";")):
            c = cls(line)
            if c is None:
                prev, run = None, 0
            else:
                put = False
                if mode == "sf":
                    put = prev == "S" and c == "F"
                elif mode == "both":
                    put = prev is not None and prev != c
                elif mode == "pair":
                    run += 1
                    put = prev is not None and run % 2 == 1 and run > 1
                if put:
                    out.append("\ts_nop 0\n")
                    added += 1
                prev = c
        out.append(line)
    open(dst, "w").writelines(out)
    print(f"{added} s_nop 0 added ({mode}; kernels {names})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
