#!/usr/bin/env python3
"""Moves the line citations of oracle/AUDIT.md (O: oracle, A: binding, L: library) along with the files they point into.
    python3 tools/audit_renumber.py [<commit>]
<commit> (default: the last commit that touched oracle/AUDIT.md) is a state in which the citations were right; every
cited line is followed through difflib's matching of that version of the file against the working tree (the
citations are taken from that commit's AUDIT.md, so the tool can be run any number of times before committing).  A citation
whose line was itself changed is reported and left alone (tests/test_oracle_audit.py will point at it)."""
import difflib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"O": "oracle/orb_oracle.cpp", "A": "include/orbx/orbmatcher_adapter.hpp", "L": "orb_slam3_amd/csrc/orbx_match.hip"}


def git(*a):
    return subprocess.run(["git", *a], cwd=ROOT, capture_output=True, text=True, check=True).stdout


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else git("log", "-n", "1", "--format=%H", "--", "oracle/AUDIT.md").strip()
    audit_path = os.path.join(ROOT, "oracle", "AUDIT.md")
    # always from the committed text, so that a second run on the same working tree does not move the citations twice
    audit = git("show", f"{commit}:oracle/AUDIT.md")
    if open(audit_path, encoding="utf-8").read() != audit:
        print("note: oracle/AUDIT.md differs from its committed text; rewriting it from the committed text")
    maps = {}
    for tag, path in FILES.items():
        old = git("show", f"{commit}:{path}").split("\n")
        new = open(os.path.join(ROOT, path), encoding="utf-8").read().split("\n")
        m = {}
        for op, i0, i1, j0, _ in difflib.SequenceMatcher(None, old, new, autojunk=False).get_opcodes():
            if op == "equal":
                for k in range(i1 - i0):
                    m[i0 + k + 1] = j0 + k + 1
        maps[tag] = m
    moved, lost = 0, []

    def rep(mo):
        nonlocal moved
        tag, ln = mo.group(1), int(mo.group(2))
        to = maps[tag].get(ln)
        if to is None:
            lost.append(f"{tag}:{ln}")
            return mo.group(0)
        moved += to != ln
        return f"{tag}:{to}"

    audit = re.sub(r"\b([OAL]):(\d+)", rep, audit)
    open(audit_path, "w", encoding="utf-8").write(audit)
    print(f"against {commit[:10]}: {moved} citations moved, {len(lost)} point at changed lines {lost[:10]}")


if __name__ == "__main__":
    main()
