#!/usr/bin/env python3
"""Sums one rocprofv3 PMC counter per kernel name: ``aggregate_pmc.py <counter_collection.csv> <COUNTER>``.
Output lines ``<kernel name>  launches=N total=T per_launch=T/N`` (the format bench.py's pmc_traffic_bytes reads)."""
import csv
import sys
from collections import defaultdict


def main():
    path, counter = sys.argv[1], sys.argv[2]
    tot, launches = defaultdict(float), defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            launches[name].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    for name in sorted(tot, key=lambda k: -tot[k]):
        n = max(1, len(launches[name]))
        print(f"{name[:60]:60s}   launches={n} total={tot[name]:.1f} per_launch={tot[name] / n:.3f}")


if __name__ == "__main__":
    main()
