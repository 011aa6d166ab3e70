#!/usr/bin/env python3
"""Median of every counter of one rocprofv3 --pmc pass for kernels whose name contains a substring.
    python3 tools/pmc_kernel.py counter_collection.csv k_spmv_tile
"""
import csv
import statistics
import sys

per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print(f"{k:28s} median {statistics.median(v):16.1f}  n {len(v)}")
