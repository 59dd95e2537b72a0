#!/usr/bin/env python
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into the per-kernel table
kept under profiles/ (name, calls, average / total duration, share)."""
import csv
import glob
import re
import sys


def main(run_dir, out_path, note=""):
    stats = glob.glob(f"{run_dir}/**/*kernel_stats.csv", recursive=True)
    if not stats:
        raise SystemExit(f"no *kernel_stats.csv under {run_dir}")
    rows = list(csv.DictReader(open(stats[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({stats[0]})", f"# {note}" if note else "#",
             f"# total kernel time {tot / 1e6:.1f} ms", "kernel,calls,avg_us,min_us,max_us,total_ms,percent"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\s+", " ", name).replace(",", ";")[:120]
        lines.append(f"{name},{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},"
                     f"{float(r['MaxNs']) / 1e3:.2f},{float(r['TotalDurationNs']) / 1e6:.2f},"
                     f"{100 * float(r['TotalDurationNs']) / tot:.2f}")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
