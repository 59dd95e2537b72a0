#!/usr/bin/env python
"""Sum a rocprofv3 --pmc counter_collection.csv per (kernel, counter): mean value per dispatch.
usage: pmc_summary.py <run_dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict


def main(run_dir, sub=""):
    files = glob.glob(f"{run_dir}/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {run_dir}")
    acc = defaultdict(lambda: defaultdict(float))     # (kernel, counter) -> dispatch id -> value
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if sub and sub not in name:
                continue
            acc[(name[:60], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (name, ctr), d in sorted(acc.items()):
        vals = list(d.values())
        print(f"{name:60s} {ctr:12s} dispatches {len(vals):4d} mean {sum(vals) / len(vals):14.1f} min {min(vals):14.1f} max {max(vals):14.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
