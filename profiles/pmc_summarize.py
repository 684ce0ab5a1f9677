#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean per dispatch.
usage: python profiles/pmc_summarize.py gpurun_out/pmc/<tag>_*  ->  prints a table"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:72].replace(",", ";").replace(" ", "")


def main(dirs):
    table = defaultdict(dict)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, cs in acc.items():
                for c, v in cs.items():
                    # skip warm-up dispatches: use the last value per kernel
                    table[k][c] = v[-1]
    counters = sorted({c for k in table for c in table[k]})
    print("kernel," + ",".join(counters))
    for k in sorted(table):
        if any(x in k for x in ("at::native", "rocclr", "elementwise")):
            continue
        print(k + "," + ",".join(f"{table[k].get(c, float('nan')):.4g}" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1:])
