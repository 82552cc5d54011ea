#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of one measurement round (gpurun_out/<tag>_*) into the small
summaries kept under profiles/.  python tools/collect_profiles.py r02"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def stats(src, dst, note):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write(f"# {note}\n")
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for r in rows:
            name = r["Name"].split("(")[0].replace("void ", "")[:60]
            f.write(f"{name},{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f},"
                    f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{r['Percentage']}\n")


def pmc(paths, dst, note):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in agg.values() for c in k})
    with open(dst, "w") as f:
        f.write(f"# {note}\n# mean counter value per dispatch\n")
        f.write("kernel,dispatches," + ",".join(counters) + "\n")
        for k, v in agg.items():
            if not k.startswith("k_"):
                continue
            n = max(len(x) for x in v.values())
            f.write(k + "," + str(n) + "," + ",".join(f"{sum(v[c]) / len(v[c]):.1f}" if c in v else "" for c in counters) + "\n")
    return agg


def main(tag):
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_bench.json"))
    bench = json.load(open(os.path.join(G, f"{tag}_bench.json")))
    cfg = bench["config"]
    frames = cfg["pairs_per_batch_per_gpu"] if "pairs_per_batch_per_gpu" in cfg else cfg["pairs_per_step_per_gpu"]
    stats(os.path.join(G, f"{tag}_prof", f"{tag}_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"),
          f"rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-pairs 0 --no-host-path   ({frames} stereo pairs per batch, "
          f"{cfg.get('batches_per_step', 1)} batches per step, {bench['steps']} steps + {bench['warmup']} warm-up, plus the one-at-a-time, "
          f"event-timed and roofline passes; 24+1 kNN searches 5000 x 10M)")
    # the launches of bench.py's roofline pass inside that trace: k_fast over the 752x480 batch with no other k_fast in flight
    tr = [r for r in csv.DictReader(open(os.path.join(G, f"{tag}_prof", f"{tag}_kernel_trace.csv"))) if "k_fast" in r["Kernel_Name"]]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_Y", r.get("Grid_Size", ""))) for r in tr)
    euroc = [e - b for b, e, g in iv if g == str(frames)]
    k = 2 * bench["steps"]
    # the pass is 2 x `steps` consecutive launches of the left extractor alone: the window of that length with the smallest mean
    best = min(range(len(euroc) - k + 1), key=lambda i: sum(euroc[i:i + k]))
    win = euroc[best:best + k]
    with open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"), "a") as f:
        f.write(f"# k_fast, the {k} consecutive launches of bench.py's roofline pass ({frames} frames 752x480, the left extractor alone on "
                f"its stream; launches {best}..{best + k - 1} of {len(euroc)}): average {sum(win) / k / 1e3:.1f} us, min {min(win) / 1e3:.1f}, "
                f"max {max(win) / 1e3:.1f}; bench.py's event-timed figure for the same pass: {bench['roofline']['avg_launch_ms'] * 1e3:.1f} us\n")
    stats(os.path.join(G, f"{tag}_euroc64", "p_kernel_stats.csv"), os.path.join(P, f"{tag}_extract_euroc_b64_kernel_stats.csv"),
          "rocprofv3 --kernel-trace --stats -- python3 tools/prof_extract.py 64 5   (one stream, 64 frames 752x480 / 1000 features per launch)")
    stats(os.path.join(G, f"{tag}_1080p", "p_kernel_stats.csv"), os.path.join(P, f"{tag}_extract_1080p_b32_kernel_stats.csv"),
          "rocprofv3 --kernel-trace --stats -- python3 tools/prof_extract.py 32 4 1920 1080 5000   (one stream, 32 frames 1920x1080 / 5000 features per launch)")
    frames = 64
    a = pmc([os.path.join(G, f"{tag}_pmc_fetch", "p_counter_collection.csv"), os.path.join(G, f"{tag}_pmc_write", "p_counter_collection.csv")],
            os.path.join(P, f"{tag}_extract_euroc_b64_pmc_hbm.csv"),
            f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 tools/prof_extract.py 64 2; "
            f"unit KB; {frames} frames 752x480 per launch; FETCH_SIZE under-reports wide reads by ~2x on gfx950")
    pmc([os.path.join(G, f"{tag}_sq1", "p_counter_collection.csv"), os.path.join(G, f"{tag}_sq2", "p_counter_collection.csv")],
        os.path.join(P, f"{tag}_extract_euroc_b64_pmc_sq.csv"),
        "rocprofv3 --pmc <8 SQ counters> (two separate runs) -- python3 tools/prof_extract.py 64 2; 64 frames 752x480 per launch")
    kf = [k for k in a if k.startswith("k_fast")][0]
    fetch = sum(a[kf]["FETCH_SIZE"]) / len(a[kf]["FETCH_SIZE"]) * 1024
    write = sum(a[kf]["WRITE_SIZE"]) / len(a[kf]["WRITE_SIZE"]) * 1024
    json.dump({"kernel": "k_fast", "frames_per_launch": frames, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
               "fetch_correction": 2.0, "source": f"profiles/{tag}_extract_euroc_b64_pmc_hbm.csv"},
              open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for src, dst in [(f"{tag}_ubench_valu.txt", f"{tag}_ubench_valu_issue.txt"), ("host_rate.txt", f"{tag}_host_path_rate.txt"),
                     (f"{tag}_phase_cycles.txt", f"{tag}_kernel_phase_cycles.txt"),
                     (f"{tag}_phase_insts.txt", f"{tag}_kernel_phase_instructions.txt"),
                     (f"{tag}_stage.txt", f"{tag}_stage_kernel_rate.txt")]:
        if os.path.exists(os.path.join(G, src)):
            lines = [l for l in open(os.path.join(G, src)) if "amdgpu.ids" not in l]
            open(os.path.join(P, dst), "w").writelines(lines)
    print("fetch raw %.1f MB, write %.1f MB per launch of %d frames" % (fetch / 1e6, write / 1e6, frames))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
