"""Per-phase instruction counts of k_fast / k_describe from the phase-stop builds (tools/prof_phase_insts.sh)."""
import collections
import csv
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/phase_insts"
CTRS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"]


def read(tag, kernel):
    agg = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(os.path.join(root, tag, "p_counter_collection.csv"))):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    return {c: agg[c] / max(n[c], 1) for c in CTRS}


for kernel, pre, names in (("k_fast", "f", ["tile load", "quick reject + queue", "arc strength", "non-max suppression", "round trip + emit"]),
                           ("k_describe", "d", ["patch load", "orientation", "horizontal blur", "vertical blur", "steered tests + output"])):
    rows = [read(f"{pre}{k}", kernel) for k in (1, 2, 3, 4)] + [read("p0", kernel)]
    waves = rows[-1]["SQ_WAVES"]
    print(f"{kernel}: {waves:.0f} waves per launch (64 frames, 752x480); wave-instructions per launch, millions (per wave)")
    print(f"  {'phase':26s}" + "".join(f"{c[9:]:>18s}" for c in CTRS[:-1]))
    prev = {c: 0.0 for c in CTRS}
    for name, r in zip(names, rows):
        print(f"  {name:26s}" + "".join(f"{(r[c] - prev[c]) / 1e6:9.2f} ({(r[c] - prev[c]) / waves:6.1f})" for c in CTRS[:-1]))
        prev = r
    print(f"  {'total':26s}" + "".join(f"{prev[c] / 1e6:9.2f} ({prev[c] / waves:6.1f})" for c in CTRS[:-1]))
