"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel means per launch.

usage: python tools/pmc_summary.py out.json FETCH_SIZE.csv WRITE_SIZE.csv ...
Kernel names are shortened to the function name with template arguments.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else name


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                e = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                e[0] += float(row["Counter_Value"])
                e[1] += 1
    res = {k: {c: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in sorted(acc.items())}
    json.dump(res, open(out, "w"), indent=1)
    for k, cs in res.items():
        print(k, {c: round(v["mean_per_launch"]) for c, v in cs.items()})


if __name__ == "__main__":
    main()
