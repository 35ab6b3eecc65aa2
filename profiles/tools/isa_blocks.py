"""Basic blocks of one kernel from hipcc -S output with their instruction counts by class: which block is the hot loop.
usage: isa_blocks.py file.s kernel_name_substring [min_valu]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
name = sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r"^(\S*%s\S*):" % re.escape(name), text, re.M)
body = text[m.end():text.index("s_endpgm", m.end())]
blocks, label, lines = [], "entry", []
for line in body.splitlines():
    line = line.split(";")[0].strip()
    if not line or (line.startswith(".") and not line.startswith(".LBB")):
        continue
    if line.endswith(":") and line.startswith(".LBB"):
        blocks.append((label, lines))
        label, lines = line[:-1], []
        continue
    lines.append(line)
blocks.append((label, lines))
total = Counter()
for label, lines in blocks:
    c = Counter()
    for line in lines:
        op = line.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        c[kind] += 1
        if kind == "valu" and re.search(r"_f64|_rcp_|_mul_lo_u32|_mul_hi_u32|_exp_|_log_|_rsq_|_sqrt_", op):
            c["slow"] += 1
    total += c
    branches = [l.split()[-1] for l in lines if l.startswith(("s_cbranch", "s_branch"))]
    if c["valu"] >= least:
        print("%-10s valu %4d (slow %3d) salu %4d lds %3d vmem %3d  -> %s" % (label, c["valu"], c["slow"], c["salu"], c["lds"], c["vmem"], " ".join(branches)))
print("total", dict(total))
