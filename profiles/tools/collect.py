#!/usr/bin/env python3
"""After `gpurun -- bash profiles/tools/refresh.sh rNN` (+ pmc_sq.sh > gpurun_out/rNN_pmc_sq.txt): copy the summaries gpurun
merged into gpurun_out/ over to profiles/ and stamp the traffic summary with the commit (the GPU box holds no .git).
usage: collect.py rNN"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
round_name = sys.argv[1]
scratch = os.path.join(ROOT, "gpurun_out")
for path in glob.glob(os.path.join(scratch, round_name + "_*_kernel_stats.csv")) + glob.glob(os.path.join(scratch, round_name + "_bench_*.json")):
    shutil.copy(path, os.path.join(ROOT, "profiles", os.path.basename(path)))
rates = os.path.join(scratch, round_name + "_fastq_to_sam_rates.txt")
if os.path.exists(rates):
    shutil.copy(rates, os.path.join(ROOT, "profiles", os.path.basename(rates)))
summary = json.load(open(os.path.join(scratch, round_name + "_traffic.json")))
summary["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "pheniqs_amd/csrc", "include"]).decode().strip()
summary["tree"] = "clean" if not dirty else "sources differ from the commit"
json.dump(summary, open(os.path.join(ROOT, "profiles", round_name + "_traffic.json"), "w"), indent=1)
counters = os.path.join(scratch, round_name + "_pmc_sq.txt")
if os.path.exists(counters):
    names = {"stream": "pamld_stream_kernel C2 100M reads", "exact": "pamld_cross_exact_kernel C2", "near": "pamld_near_kernel C2", "lut": "pamld_lut_kernel (consumer) C2"}
    with open(os.path.join(ROOT, "profiles", round_name + "_fast_pmc_sq.csv"), "w") as out:
        out.write("counter,average_per_launch,launches,kernel\n")
        for line in open(counters):
            m = re.match(r"gpurun_out/pmc_\d+/ (\w+) (\w+) ([\d.]+) (\d+)", line)
            if m:
                out.write("%s,%s,%s,%s\n" % (m.group(2), m.group(3), m.group(4), names.get(m.group(1), m.group(1))))
print(summary["build"][:16], summary["commit"], summary["tree"])
