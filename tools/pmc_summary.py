#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import collections, csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    print(path)
    for k, v in agg.items():
        print("  %-28s" % k, {c: (round(sum(x) / len(x), 1), len(x)) for c, x in v.items()})
