"""The host staged leg of bench.py alone: python profiles/tools/scratch/staged.py [reads]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import bench
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
torch.cuda.set_device(0)
w = bench.Workload(torch, torch.device("cuda", 0), 0, 0, "C2", "pamld", "fast", 1 << 20, 0)
out = bench.host_staged(torch, w, reads)
print(json.dumps(out))
