"""Per-kernel counter averages from tools/pmc_kernels.sh: python3 tools/pmc_table.py gpurun_out/<tag>"""
import collections
import csv
import sys
pre = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for suf in ("_a", "_b"):
    for r in csv.DictReader(open(f"{pre}{suf}/p_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"] + "#" + r["Dispatch_Id"]] = 1
for k, v in agg.items():
    if not k.startswith("k_"):
        continue
    launches = {}
    for key in cnt[k]:
        c = key.split("#")[0]
        launches[c] = launches.get(c, 0) + 1
    row = {c: x / launches[c] / 1e6 for c, x in v.items()}
    lds = row.get("SQ_ACTIVE_INST_LDS", 0)
    print(f"{k:14s} VALU {row.get('SQ_INSTS_VALU', 0):7.2f}M SALU {row.get('SQ_INSTS_SALU', 0):6.2f}M LDS {row.get('SQ_INSTS_LDS', 0):5.2f}M "
          f"ldsActive {lds:6.2f}M conflict {row.get('SQ_LDS_BANK_CONFLICT', 0):6.2f}M ({row.get('SQ_LDS_BANK_CONFLICT', 0) / lds if lds else 0:4.2f}) "
          f"waveCyc {row.get('SQ_WAVE_CYCLES', 0):7.1f}M waitAny {row.get('SQ_WAIT_ANY', 0):7.1f}M waitInst {row.get('SQ_WAIT_INST_ANY', 0):7.1f}M busy {row.get('SQ_BUSY_CYCLES', 0):5.2f}M")
