"""C2 at small batches: the chain's kernels with their HIP event times, and the wall clock per call (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
import torch, shape_grid
from pheniqs_amd import synthetic
for config in sys.argv[1:] or ["C2"]:
    for reads in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
        rate, ms, chain, _ = shape_grid.run(torch, synthetic.config_spec(config, "pamld"), reads, steps=50, warmup=5)
        print(config, reads, "%.3f ms/call %.2f Greads/s" % (ms, rate), [(k.replace("pamld_", "").replace("_kernel", ""), round(t * 1e3, 1)) for k, t in chain], "us; sum %.1f us" % (sum(t for _, t in chain) * 1e3), flush=True)
