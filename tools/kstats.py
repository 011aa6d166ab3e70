#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *_kernel_stats.csv."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f'{r["Name"][:58]:58s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.2f} '
          f'tot_ms {float(r["TotalDurationNs"]) / 1e6:8.2f}')
