#!/usr/bin/env python3
"""Fold the rocprofv3 passes of profiles/tools/refresh.sh into profiles/rNN_traffic.json, the file bench.py quotes
`roofline.traffic` from.

usage: summarize.py <gpurun_out/prof_rNN> <out.json> shape [shape ...]

Per shape directory: trace/ (rocprofv3 --kernel-trace --stats --output-format csv), fetch/ and write/ (rocprofv3 --pmc
FETCH_SIZE / --pmc WRITE_SIZE, passes of their own) and bench.json (the bench line of the trace pass).  HBM bytes follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and doubled on gfx950 (128 byte requests are
tallied as 64), WRITE_SIZE is in KiB and taken as is.  A shape's `step_bytes_per_read` sums EVERY kernel of the library a
step launches; `wasted` = that over the algorithmic bytes.  The file names the build the passes ran on
(phq_build_id, the SHA-256 over the library's sources) and the commit: bench.py flags a summary of another build as stale.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def newest(directory, pattern):
    found = sorted(glob.glob(os.path.join(directory, "**", pattern), recursive=True), key=os.path.getmtime)
    if not found:
        raise SystemExit("no %s under %s" % (pattern, directory))
    return found[-1]


def counter_by_kernel(directory, counter):
    """{kernel: (summed value over its dispatches, dispatches)}"""
    total = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for row in csv.DictReader(open(newest(directory, "*_counter_collection.csv"))):
        if row["Counter_Name"] == counter and "phq::" in row["Kernel_Name"] and "copy_probe" not in row["Kernel_Name"]:      # the copy probe (bench.py's measured_copy_GBs) is no kernel of a step
            total[row["Kernel_Name"]] += float(row["Counter_Value"])
            calls[row["Kernel_Name"]] += 1
    return {name: (total[name], calls[name]) for name in total}


def short(name):
    return name.replace("void phq::", "").replace("(phq::StageArgs)", "")


def main():
    directory, out = sys.argv[1:3]
    from pheniqs_amd import capi
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = None                                   # the GPU box holds no .git: the caller may add it
    summary = {"build": capi.build_id(), "source": capi.source_id(), "commit": commit, "shapes": {},
               "note": "traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (gfx950 tallies 128 B requests as 64 B); "
                       "per launch averages over every dispatch of a pass; the steps of a pass (timed + warmup) are read from the bench line it printed"}
    for shape in sys.argv[3:]:
        where = os.path.join(directory, shape)
        line = json.load(open(os.path.join(where, "bench.json")))
        reads = line["config"]["reads_per_gpu"]
        algorithmic = line["roofline"]["algorithmic_bytes_per_read"]
        fetch = counter_by_kernel(os.path.join(where, "fetch"), "FETCH_SIZE")
        write = counter_by_kernel(os.path.join(where, "write"), "WRITE_SIZE")
        def steps_of(name):
            """timed + warmup steps of a pass, from the bench line that pass printed (refresh.sh keeps all three)"""
            path = os.path.join(where, name)
            try:
                pass_line = json.load(open(path))
                return int(pass_line["steps"]) + int(pass_line["warmup"])
            except (OSError, ValueError, KeyError):
                raise SystemExit("%s: no bench line with steps and warmup; the traffic of %s cannot be scaled" % (path, shape))
        trace_steps, fetch_steps, write_steps = steps_of("bench.json"), steps_of("fetch.json"), steps_of("write.json")
        kernels = {}
        for row in csv.DictReader(open(newest(os.path.join(where, "trace"), "*_kernel_stats.csv"))):
            name = row["Name"]
            if "phq::" not in name or "copy_probe" in name:
                continue
            entry = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "launches_per_step": int(row["Calls"]) / float(trace_steps)}
            if name in fetch and name in write:
                entry["fetch_kib_per_launch"] = fetch[name][0] / fetch[name][1]
                entry["write_kib_per_launch"] = write[name][0] / write[name][1]
                entry["bytes_per_step"] = (2.0 * fetch[name][0] / fetch_steps + write[name][0] / write_steps) * 1024.0
                entry["bytes_per_read"] = entry["bytes_per_step"] / reads
            kernels[short(name)] = entry
        step = sum(k.get("bytes_per_step", 0.0) for k in kernels.values())
        dominant = max(kernels, key=lambda k: kernels[k]["average_ns"] * kernels[k]["launches_per_step"])
        # bench.py's dominant kernel is a KIND (all launches of it in a step): every kernel of that family together
        family = dominant.split("<")[0]
        dominant_bytes = sum(k.get("bytes_per_step", 0.0) for n, k in kernels.items() if n.split("<")[0] == family)
        summary["shapes"][shape] = {"reads_per_step": reads, "algorithmic_bytes_per_read": algorithmic, "ms_per_step": line["ms_per_step"],
                                    "dominant_kernel": family, "dominant_bytes_per_read": dominant_bytes / reads,
                                    "step_bytes_per_read": step / reads, "wasted": step / reads / algorithmic, "kernels": kernels}
    json.dump(summary, open(out, "w"), indent=1)
    for shape, entry in summary["shapes"].items():
        print("%-28s %6.3f ms  step %.2f B/read over %d algorithmic = %.2fx  (dominant %s %.2f B/read)" % (
            shape, entry["ms_per_step"], entry["step_bytes_per_read"], entry["algorithmic_bytes_per_read"], entry["wasted"], entry["dominant_kernel"], entry["dominant_bytes_per_read"]))


if __name__ == "__main__":
    main()
