#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv
import collections
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(rows.items()):
    n = max(len(v) for v in ctrs.values())
    print("%s  (%d dispatches)" % (name[:90], n))
    for c, v in sorted(ctrs.items()):
        print("    %-28s %16.0f" % (c, sum(v) / len(v)))
