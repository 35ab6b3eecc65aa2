#!/usr/bin/env python3
"""Fold rocprofv3 output into the per kernel summary bench.py quotes (profiles/rNN_summary.json).

usage: summarize.py <kernel-trace dir> <FETCH_SIZE pmc dir> <WRITE_SIZE pmc dir> <reads per launch> <out.json> [note]

  kernel-trace dir : rocprofv3 --kernel-trace --stats --output-format csv   (*_kernel_stats.csv)
  pmc dirs         : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE           (*_counter_collection.csv), own passes

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and is
doubled on gfx950 (128 byte requests are tallied as 64), WRITE_SIZE is in KiB and taken as is.
"""
import collections
import csv
import glob
import json
import os
import sys

KINDS = (("pamld_near", ("pamld_near_kernel",)),
         ("pamld_product", ("pamld_lut_kernel", "pamld_cross_kernel", "pamld_wide_kernel")),
         ("pamld_exact", ("pamld_cross_exact_kernel",)),
         ("pamld_literal", ("pamld_literal_kernel",)),
         ("mdd_exact", ("mdd_exact_kernel",)),
         ("mdd", ("mdd_scan_kernel", "mdd_kernel")),
         ("count", ("count_kernel",)))


def kind_of(name):
    for kind, needles in KINDS:
        if any(n in name for n in needles):
            return kind
    return None


def first(directory, pattern):
    """The newest match: a directory may hold the output of earlier runs too."""
    found = sorted(glob.glob(os.path.join(directory, "**", pattern), recursive=True), key=os.path.getmtime)
    if not found:
        raise SystemExit("no %s under %s" % (pattern, directory))
    return found[-1]


def counter_average(directory, counter):
    total = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for row in csv.DictReader(open(first(directory, "*_counter_collection.csv"))):
        if row["Counter_Name"] == counter:
            total[row["Kernel_Name"]] += float(row["Counter_Value"])
            calls[row["Kernel_Name"]] += 1
    return {name: total[name] / calls[name] for name in total}


def main():
    trace, fetch_dir, write_dir, reads, out = sys.argv[1:6]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    reads = int(reads)
    fetch = counter_average(fetch_dir, "FETCH_SIZE")
    write = counter_average(write_dir, "WRITE_SIZE")
    summary = {}
    for row in csv.DictReader(open(first(trace, "*_kernel_stats.csv"))):
        name = row["Name"]
        kind = kind_of(name)
        if kind is None or kind in summary:
            continue
        entry = {"kernel": name, "calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "reads_per_launch": reads}
        if name in fetch and name in write:
            traffic = (2.0 * fetch[name] + write[name]) * 1024.0
            entry.update({"fetch_size_kib": fetch[name], "write_size_kib": write[name], "traffic_bytes_per_launch": traffic,
                          "traffic_bytes_per_read": traffic / reads,
                          "note": ("FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 counts 128 B requests as 64 B). " + note).strip()})
        summary[kind] = entry
    json.dump(summary, open(out, "w"), indent=2)
    print(json.dumps(summary, indent=2))


if __name__ == "__main__":
    main()
