#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, average/min/max duration.
The stock kernel_stats.csv averages one kernel over launches of different shapes (wires / zs / quotient
batches); this keeps them apart so each line can be compared with bench.py's per-stage hipEvent times."""
import csv
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            grid = "%sx%sx%s" % (row["Grid_Size_X"], row["Grid_Size_Y"], row["Grid_Size_Z"])
            acc[(name, grid)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    print("%-44s %-20s %6s %12s %12s %12s %12s" % ("kernel", "grid(threads)", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"))
    for (name, grid), v in rows:
        print("%-44s %-20s %6d %12.3f %12.3f %12.3f %12.3f" % (name[:44], grid, len(v), sum(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1])
