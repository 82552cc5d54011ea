#!/usr/bin/env python3
"""Static instruction mix of k_fast and k_describe per phase, from an assembly listing with phase markers:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DORBX_ISA_MARKERS --cuda-device-only -S \
          -o build/asm/extract_markers.s orb_slam3_amd/csrc/orbx_extract.hip
    python3 tools/isa_phase_table.py build/asm/extract_markers.s
Counts are instructions in the listing (a loop body counts once), split by what they cost the issue stage: vector ALU,
vector memory, LDS, scalar memory, and the scalar side by kind -- waits, no-ops, branches, exec-mask bookkeeping,
compares / selects, and plain scalar arithmetic (address and index arithmetic for the most part)."""
import collections
import json
import re
import sys

KERNELS = {"fast": ("k_fastILi48", ["tile load", "quick reject + queue", "arc strength", "non-max suppression", "round trip + emit"]),
           "describe": ("k_describePK", ["patch load", "orientation", "horizontal blur", "vertical blur", "steered tests + output"])}


def kind(m):
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith("ds_"): return "lds"
    if m.startswith("v_"): return "valu"
    if m.startswith(("s_load", "s_buffer_load")): return "smem"
    if m == "s_waitcnt": return "s_waitcnt"
    if m.startswith("s_nop"): return "s_nop"
    if m.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")): return "branch/barrier"
    if "saveexec" in m or m in ("s_or_b64", "s_and_b64", "s_andn2_b64", "s_xor_b64", "s_mov_b64", "s_not_b64") : return "exec/mask"
    if m.startswith(("s_cmp", "s_cselect", "s_bitcmp")): return "cmp/select"
    if m.startswith("s_"): return "scalar alu"
    return "other"


# Measured issue cost of vector instructions, ns per wave-instruction and SIMD with 8 waves per SIMD (tools/ubench/valu_rates.hip,
# profiles/r02_ubench_valu_classes.txt, profiles/r03_ubench_f64_rates.txt): the plain 32-bit moves / logic / adds / float add and
# multiply issue every other cycle, nearly everything else every fourth.
VALU_COST_NS = {"v_mov_b32": 0.93, "v_and_b32": 0.95, "v_or_b32": 0.94, "v_xor_b32": 0.87, "v_not_b32": 0.92, "v_add_u32": 0.87,
                "v_sub_u32": 0.86, "v_subrev_u32": 0.86, "v_lshrrev_b32": 0.88, "v_add_f32": 0.95, "v_mul_f32": 0.97, "v_sub_f32": 0.95,
                "v_fma_f32": 1.21, "v_fmac_f32": 1.21, "v_bitop3_b32": 1.16, "v_rcp_f32": 3.47, "v_rcp_iflag_f32": 3.47}
VALU_COST_DEFAULT_NS = 1.78


def valu_cost(m):
    m = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    return VALU_COST_NS.get(m, VALU_COST_DEFAULT_NS)


COLS = ["valu", "vmem", "lds", "smem", "s_waitcnt", "s_nop", "branch/barrier", "exec/mask", "cmp/select", "scalar alu"]


def whole_kernel_mix(text):
    """Per kernel of the listing: vector instructions and their mean issue cost by the table above (static mix)."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", text, re.S | re.M):
        costs = [valu_cost(l.strip().split()[0]) for l in m.group(2).split("\n")
                 if l.startswith("\tv_")]
        if costs:
            out[m.group(1)] = {"valu_in_listing": len(costs), "mean_valu_cost_ns": sum(costs) / len(costs)}
    return out


def main(path):
    text = open(path).read()
    if len(sys.argv) > 2 and sys.argv[2] == "--json":
        print(json.dumps(whole_kernel_mix(text), indent=1))
        return
    for name, (sym, phases) in KERNELS.items():
        m = re.search(r"^(_Z\w*%s\w*):.*?\n(.*?)\.Lfunc_end" % sym, text, re.S | re.M)
        if not m:
            print(name, "not found"); continue
        counts = [collections.Counter() for _ in phases]
        ph = 0
        for line in m.group(2).split("\n"):
            t = line.strip()
            mk = re.match(r"; orbx-phase-end %s (\d+)" % name, t)
            if mk:
                ph = min(int(mk.group(1)), len(phases) - 1)
                continue
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            counts[ph][kind(t.split()[0])] += 1
        print("%s (%s): instructions in the listing per phase" % (name, m.group(1)[:40]))
        print("  %-24s" % "phase" + "".join("%15s" % c for c in COLS) + "%15s" % "scalar/valu")
        tot = collections.Counter()
        for p, c in zip(phases, counts):
            tot.update(c)
            sc = sum(c[k] for k in COLS[4:])
            print("  %-24s" % p + "".join("%15d" % c[k] for k in COLS) + "%15.2f" % (sc / max(c["valu"], 1)))
        sc = sum(tot[k] for k in COLS[4:])
        print("  %-24s" % "total" + "".join("%15d" % tot[k] for k in COLS) + "%15.2f" % (sc / max(tot["valu"], 1)))


if __name__ == "__main__":
    main(sys.argv[1])
