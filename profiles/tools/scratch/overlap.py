"""Two half batches on two streams against one whole batch on one: does the hardware overlap the tiers of one with the other?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import bench

device = torch.device("cuda", 0)
torch.cuda.set_device(0)

def run(workloads, streams, steps=8, warmup=3):
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for w, s in zip(workloads, streams):
            w.stream = s.cuda_stream
            w.set.stats_reset(w.stream)
            w.set.classify_device(w.n, 0, w.n, w.base, w.qual, w.flags, w.result, w.word, w.stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for config, reads in (("C2", 100_000_000), ("C5", 50_000_000)):
    whole = bench.Workload(torch, device, 0, 0, config, "pamld", "fast", reads, 1 << 22)
    t_whole = run([whole], [torch.cuda.Stream()])
    whole.close()
    parts = 2
    halves = [bench.Workload(torch, device, 0, 0, config, "pamld", "fast", reads // parts, 1 << 22) for _ in range(parts)]
    t_serial = run(halves, [torch.cuda.Stream()] * parts)
    t_overlap = run(halves, [torch.cuda.Stream() for _ in range(parts)])
    for h in halves: h.close()
    print(config, "whole %.3f ms, halves one stream %.3f ms, halves two streams %.3f ms" % (t_whole, t_serial, t_overlap), flush=True)
