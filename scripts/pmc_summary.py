#!/usr/bin/env python3
"""Mean of every collected counter per kernel from rocprofv3 --pmc output (development tool).

    python scripts/pmc_summary.py <dir> [<dir> ...]
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("codd::", "").replace("(anonymous namespace)::", "")
    return name.strip()


def main():
    acc = collections.defaultdict(list)
    for root in sys.argv[1:]:
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:34s} {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main()
