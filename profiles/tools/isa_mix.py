"""Vector instruction mix of one kernel's main loop from hipcc -S output: counts by issue class (gfx950).

full rate (measured, profiles/tools/micro): v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32, v_mov_b32, v_bitop3_b32,
v_add/mul/fma/fmac/sub_f32; everything else vector is half rate.  usage: isa_mix.py file.s kernel_name_substring
"""
import re
import sys
from collections import Counter

FULL = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32", "v_bitop3_b32",
        "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_sub_f32", "v_subrev_f32", "v_not_b32", "v_add_co_u32", "v_xnor_b32"}
text = open(sys.argv[1]).read()
name = sys.argv[2]
start = None
for m in re.finditer(r"^(\S*%s\S*):" % re.escape(name), text, re.M):
    start = m.end()
    break
end = text.index(".end_amdhsa_kernel", start) if ".end_amdhsa_kernel" in text[start:] else len(text)
end = text.index("s_endpgm", start)
body = text[start:end]
# the main loop: the largest block between a label and a backward branch to it
counts = Counter()
for line in body.splitlines():
    line = line.strip()
    if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
        continue
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    counts[base] += 1
valu = {k: v for k, v in counts.items() if k.startswith("v_")}
full = sum(v for k, v in valu.items() if k in FULL)
half = sum(v for k, v in valu.items() if k not in FULL)
print("whole kernel: VALU %d (full rate %d, half rate %d) => %.0f half-rate units; SALU %d, LDS %d, VMEM %d" % (
    full + half, full, half, half + 0.62 * full, sum(v for k, v in counts.items() if k.startswith("s_")),
    sum(v for k, v in counts.items() if k.startswith("ds_")), sum(v for k, v in counts.items() if k.startswith(("global_", "buffer_", "flat_")))))
for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:40]:
    print("  %-22s %4d %s" % (k, v, "" if k in FULL else "half"))
for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
    if k.startswith(("ds_", "global_", "buffer_")):
        print("  %-22s %4d" % (k, v))
