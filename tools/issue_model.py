#!/usr/bin/env python3
"""Writes profiles/<tag>_issue_model.json: what bench.py needs to turn the committed instruction counters of the
extraction kernels into an issue-bound ceiling.
    python3 tools/issue_model.py r03 build/asm/extract_markers.s
Per kernel: the mean issue cost of its vector instructions (static mix of the listing x the measured cost of each
instruction class, tools/isa_phase_table.py) and the measured marginal cost of a scalar instruction
(profiles/r03_scalar_issue.txt).  The ceiling charges every vector and scalar wave-instruction its cost on the SIMD that
issues it and assumes all 1024 SIMDs busy all the time; memory instructions, waits and barriers are not charged."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_phase_table as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, listing):
    mix = T.whole_kernel_mix(open(listing).read())
    short = {}
    for sym, v in mix.items():
        m = re.match(r"_Z\d+(k_[a-z_0-9]+?)(ILi(\d+)E)?(EvP|P)", sym)
        if not m:
            continue
        name = m.group(1) + (f"<{m.group(3)}>" if m.group(3) else "")
        short[name] = v
    out = {"simds": 1024,
           "valu_cost_ns_by_instruction": T.VALU_COST_NS, "valu_cost_ns_default": T.VALU_COST_DEFAULT_NS,
           "salu_cost_ns": 0.76,
           "kernels": short,
           "sources": ["profiles/r02_ubench_valu_classes.txt", "profiles/r03_ubench_f64_rates.txt", "profiles/r03_scalar_issue.txt",
                       "static instruction mix of the product kernels' listing (a loop body counts once)"]}
    path = os.path.join(ROOT, "profiles", f"{tag}_issue_model.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, {k: round(v["mean_valu_cost_ns"], 3) for k, v in short.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
