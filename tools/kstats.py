"""Prints a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py gpurun_out/<dir>/p_kernel_stats.csv ..."""
import csv
import sys
for path in sys.argv[1:]:
    print(path)
    for r in csv.DictReader(open(path)):
        print(f"  {r['Name'][:44]:44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} min_us={float(r['MinNs']) / 1e3:9.1f} pct={r['Percentage']}")
