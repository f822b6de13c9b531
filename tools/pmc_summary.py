#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: sum of every counter over
all dispatches of the kernel, plus dispatch count.  Usage: pmc_summary.py DIR [substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                if filt and filt not in k:
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print(f"  {c:40s} {acc[k][c]:.6g}   (dispatches {len(calls[(k, c)])})")


if __name__ == "__main__":
    main()
