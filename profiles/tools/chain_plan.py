"""The table decoder shape -> kernel chain, generated WITHOUT a GPU from the stage dispatcher itself (phq_plan_chain: launch_stage
walks its predicates with its launches switched off): the named workloads of bench.py, then the shape grid of shape_grid.py.
    python profiles/tools/chain_plan.py > profiles/r06_chain_plan.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shape_grid
from pheniqs_amd import DecoderSet, capi, synthetic


def short(names):
    return " -> ".join(name.replace("_kernel", "") for name in names)


print("# shape -> kernel chain, as the stage dispatcher decides it (`phq_plan_chain`, no GPU; build `%s`)\n" % capi.build_id()[:16])
print("Fast mode, 100 M reads a batch.  The first kernel is the tier every read goes through; what it cannot certify goes down the chain,")
print("which always ends in the reference's own sums (`pamld_literal`).  `tests/test_shape_grid_plan.py` holds every grid shape to a tier 0")
print("first kernel or to the allow list of `shape_grid.py`; `tests/test_gpu_shape_grid.py` holds what runs on the device to this plan.\n")
print("## bench.py's named workloads\n\n| workload | algorithm | decoder | chain |\n|---|---|---|---|")
for name in ("C2", "C3", "C4", "C4V", "C5", "H94", "H94A", "S384", "S192", "U96", "U384", "L12", "L16"):
    for algorithm in ("pamld", "mdd"):
        job, _ = synthetic.config_job(name, algorithm)
        s = DecoderSet(job, device=capi.DEVICE_NONE)
        chain = s.plan_chain(100_000_000, capi.MODE_FAST)
        for k in sorted({c["decoder"] for c in chain}):
            print("| %s | %s | %d | %s |" % (name, algorithm, k, short([c["kernel"] for c in chain if c["decoder"] == k])))
        s.close()
print("\n## the shape grid (PAMLD; `L` nucleotides a segment, `N` barcodes)\n\n| family | L | N | priors | chain | no tier 0 because |\n|---|---|---|---|---|---|")
first = {}
for family, L, N, per, priors in shape_grid.shapes():
    spec, barcodes = shape_grid.spec_of(family, L, N, per, priors)
    names = shape_grid.plan(spec)
    tier0 = any(names[0].startswith(name) for name in shape_grid.TIER0)
    first[names[0]] = first.get(names[0], 0) + 1
    print("| %s | %d | %d | %s | %s | %s |" % (family, L, barcodes, priors, short(names), "" if tier0 else (shape_grid.allowed(family, L, N, priors) or "?")[:90]))
print("\nfirst kernels over the grid: " + ", ".join("%s %d" % (k.replace("_kernel", ""), v) for k, v in sorted(first.items(), key=lambda kv: -kv[1])))
