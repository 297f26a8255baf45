#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum and per-launch mean of every
counter.  HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies
128-byte read requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact."""
import csv
import sys
from collections import defaultdict


def main(paths):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0][-60:]
                c = row["Counter_Name"]
                agg[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    print("kernel,counter,launches,sum,per_launch")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            n = calls[k][c]
            print(f"{k},{c},{n},{agg[k][c]:.6g},{agg[k][c] / n:.6g}")
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            n = calls[k]["FETCH_SIZE"]
            kib = 2 * agg[k]["FETCH_SIZE"] + agg[k]["WRITE_SIZE"]
            print(f"{k},HBM_BYTES_corrected,{n},{kib * 1024:.6g},{kib * 1024 / n:.6g}")


if __name__ == "__main__":
    main(sys.argv[1:])
