#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE (KB per dispatch) per kernel from two rocprofv3 --pmc passes.
    python3 tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json
"""
import csv
import json
import re
import statistics
import sys

out = {}
for path in sys.argv[1:3]:
    per = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        per.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (name, ctr), vals in per.items():
        d = out.setdefault(name, {})
        d[ctr + "_KB_median"] = statistics.median(vals)
        d[ctr + "_n"] = len(vals)
json.dump(dict(sorted(out.items())), open(sys.argv[3], "w"), indent=1)
