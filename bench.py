#!/usr/bin/env python3
"""Benchmark of the barcode classification hot path on MI355X (contract: see the task's bench.py section).

Workload (BASELINE.json configs[1], "C2"): 100 M synthetic reads, one 8 bp index, 96 barcodes, PAMLD, all
inputs resident in HBM before the timed region.  One step = one pass of the hot path over the whole batch
(TranscodingDecoder::classify for every read + the end of run counter merge).  With N GPUs every rank
decodes its own 100 M reads (weak scaling, no data path collective) and the integer counter buffer is merged
by one RCCL all-reduce per step.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(job, code, quality, flag, target_seconds=12.0):
    """The CPU oracle (a port of the reference algorithm, oracle/pheniqs_oracle.c) timed on this host's cores:
    one private decoder set per thread, counters merged at the end, like the reference's TranscodingThreads."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    probe = min(20000, code.shape[0])
    engine = pyoracle.OracleJob(job)
    t0 = time.perf_counter()
    engine.classify(code[:probe], quality[:probe], flag[:probe])
    single = probe / (time.perf_counter() - t0)
    sample = int(min(code.shape[0], max(probe, single * cores * target_seconds)))
    bounds = np.linspace(0, sample, cores + 1).astype(np.int64)
    engines = [pyoracle.OracleJob(job) for _ in range(cores)]

    def work(k):
        a, b = bounds[k], bounds[k + 1]
        return engines[k].classify(code[a:b], quality[a:b], flag[a:b])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        results = list(pool.map(work, range(cores)))
    elapsed = time.perf_counter() - t0
    for other in engines[1:]:
        engines[0].collect(other)
    return {"value": sample / elapsed / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "port",
            "sample": "%d reads of the same C2 pool, %d threads x private decoder set, %.1f s; 1 thread: %.3f Mreads/s" % (
                sample, cores, elapsed, single / 1e6)}, results, bounds


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmup", type=int, default=3)
    parser.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU per step")
    parser.add_argument("--pool", type=int, default=1 << 20, help="distinct synthetic reads tiled to --reads")
    parser.add_argument("--algorithm", default="pamld", choices=["pamld", "mdd"])
    parser.add_argument("--mode", default="literal", choices=["literal", "fast"])
    parser.add_argument("--no-cpu-baseline", action="store_true")
    args = parser.parse_args()

    import torch
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, distributed, synthetic

    rank, local_rank, world = distributed.initialise("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    job, recipe = synthetic.config_job("C2", args.algorithm)
    bytes_per_read = synthetic.algorithmic_bytes_per_read(job)          # 37 for C2
    pool_n = min(args.pool, args.reads)
    tiles = -(-args.reads // pool_n)
    n = args.reads
    code, quality, flag, offsets = synthetic.generate_rows(job, pool_n, recipe["seed"] + 1000 * rank)

    s = DecoderSet(job, device=local_rank)
    s.set_mode(capi.MODE_FAST if args.mode == "fast" else capi.MODE_LITERAL)
    stride = s.decoders[0]["plane_stride"]
    base = torch.from_numpy(s.planes_from_rows(code, offsets)).to(device).view(pool_n, stride).repeat(tiles, 1)[:n].contiguous().view(-1)
    qual = torch.from_numpy(s.planes_from_rows(quality, offsets)).to(device).view(pool_n, stride).repeat(tiles, 1)[:n].contiguous().view(-1)
    flags = torch.from_numpy(flag).to(device).repeat(tiles)[:n].contiguous()
    result = torch.zeros(n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device)
    word = torch.zeros(n, dtype=torch.int32, device=device)
    counters = s.stats_tensor()
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step(events=None):
        s.stats_reset(stream)
        if events is not None:
            events[0].record()
        s.classify_device(n, 0, n, base, qual, flags, result, word, stream)
        if events is not None:
            events[1].record()
        distributed.allreduce_counters(counters)

    for _ in range(args.warmup):
        step()
    barrier()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(pairs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))

    # size independent check of the last step: counters are a whole number of pool passes (+ remainder)
    words = s.stats_export()
    total = int(words[0] + words[1])
    assert total == n * world, (total, n, world)

    if rank == 0:
        line = {
            "metric": "Mreads/s decoded (PAMLD, 96-plex 8bp)" if args.algorithm == "pamld" else "Mreads/s decoded (MDD, 96-plex 8bp)",
            "value": n * world * args.steps / elapsed / 1e6,
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.algorithm == "pamld" else "u32",
            "data": "synthetic",
            "config": {"workload": "C2: %d reads per GPU, single 8 bp index, 96 barcodes, %s, inputs resident in HBM" % (n, args.algorithm.upper()),
                       "reads_per_gpu": n, "pool": pool_n, "barcodes": 96, "barcode_length": 8, "kernel_mode": args.mode,
                       "sharding": "reads x%d, counters merged by one all-reduce per step" % world},
            "roofline": {"bound": "hbm", "achieved": bytes_per_read * n / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bytes_per_read * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_read": bytes_per_read},
        }
        if world == 1 and not args.no_cpu_baseline:
            baseline, cpu_results, bounds = cpu_baseline(job, code, quality, flag)
            line["cpu_baseline"] = baseline
            # the same reads through the GPU path must give the same records
            got = result[: pool_n * RESULT_DTYPE.itemsize].cpu().numpy().view(RESULT_DTYPE)
            checked = 0
            for k, (records, _) in enumerate(cpu_results):
                a, b = int(bounds[k]), int(bounds[k + 1])
                for name in ("index", "distance", "hq_distance", "status", "qcfail"):
                    assert np.array_equal(got[a:b][name], records[:, 0][name]), name
                assert np.abs(got[a:b]["confidence"] - records[:, 0]["confidence"]).max() < 1e-6
                checked += b - a
            line["parity_checked_reads"] = checked
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
