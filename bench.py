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


def profiled_traffic(mode, reads, config="C2", algorithm="pamld", kernel_mode="fast"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN_summary.json),
    scaled to this run's reads per launch.  PMC counters cannot be read from inside the benchmark itself, and the
    committed summaries are of the headline workload only (C2, PAMLD): null for anything else."""
    import glob
    if config != "C2" or algorithm != "pamld":
        return None
    pattern = "r*_summary.json" if kernel_mode == "fast" else "r*_summary_%s.json" % kernel_mode
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        if kernel_mode == "fast" and not os.path.basename(path).endswith("_summary.json"):
            continue
        try:
            entry = json.load(open(path)).get(mode)
        except (OSError, ValueError):
            continue
        if entry:
            return entry["traffic_bytes_per_read"] * reads
    return None


def measured_copy_bandwidth(torch, device):
    """the practical HBM ceiling of this node next to the 8 TB/s the roofline is priced against — a
    device to device copy of 1 GiB, bytes read + bytes written per second (torch's copy kernel, HIP events)."""
    words = 1 << 28
    source = torch.empty(words, dtype=torch.int32, device=device).fill_(1)
    target = torch.empty_like(source)
    for _ in range(2):
        target.copy_(source)
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    begin.record()
    repeats = 10
    for _ in range(repeats):
        target.copy_(source)
    end.record()
    torch.cuda.synchronize(device)
    return 2 * 4 * words * repeats / (begin.elapsed_time(end) * 1e-3) / 1e9


def fastq_end_to_end(args, code, quality, flag, n):
    """Next rows f1 + f2 end to end: synthetic FASTQ text on disk -> phq_transcode_fastq_to_sam -> SAM text
    (uncompressed, /dev/null).  One file per input segment: a 50 nt template read and the index read; the staged
    rows of the pool are written back as text, so the decoder sees the same observations as the kernel bench.
    Host bound (parse and format on the host, classification on the GPU); never the headline value."""
    import tempfile
    from pheniqs_amd import DecoderSet, capi, synthetic
    spec = synthetic.config_spec(args.config, args.algorithm)
    if args.config != "C2":
        return {"skipped": "the FASTQ leg writes the two segment C2 layout only"}
    spec["template"] = {"transform": {"token": ["0::"]}}
    s = DecoderSet(spec, device=0)
    s.set_mode(capi.MODE_FAST if args.mode == "fast" else capi.MODE_LITERAL)
    rng = np.random.default_rng(7)
    letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)                 # iupac.h BamToAmbiguousAscii
    pool = code.shape[0]
    reps = -(-n // pool)
    index_text = letters[np.tile(code, (reps, 1))[:n]]
    index_quality = (np.minimum(np.tile(quality, (reps, 1))[:n], 93) + 33).astype(np.uint8)
    template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 50))]
    template_quality = np.full((n, 50), ord("F"), dtype=np.uint8)
    filtered = np.tile(flag, reps)[:n]
    with tempfile.TemporaryDirectory() as directory:
        paths = []
        for k, (text, qual) in enumerate(((template, template_quality), (index_text, index_quality))):
            path = os.path.join(directory, "segment_%d.fastq" % k)
            with open(path, "wb") as out:
                for first in range(0, n, 1 << 16):
                    chunk = []
                    for i in range(first, min(n, first + (1 << 16))):
                        chunk.append(b"@M:1:FC:1:1101:%d:%d %d:%s:0:1\n%s\n+\n%s\n" % (
                            i >> 16, i & 0xffff, k + 1, b"Y" if filtered[i] else b"N", text[i].tobytes(), qual[i].tobytes()))
                    out.write(b"".join(chunk))
            paths.append(path)
        size = sum(os.path.getsize(p) for p in paths)
        s.transcode_fastq_to_sam(paths, "/dev/null", 1 << 20)                    # warm: allocations, page cache
        t0 = time.perf_counter()
        incoming, incoming_pf = s.transcode_fastq_to_sam(paths, "/dev/null", 1 << 20)
        dt = time.perf_counter() - t0
    assert incoming == n
    return {"reads": n, "Mreads/s": n / dt / 1e6, "fastq_MB/s": size / dt / 1e6, "seconds": dt,
            "note": "2 uncompressed FASTQ files (50 nt template + 8 nt index) -> SAM text to /dev/null; per file one reader thread indexing lines + decoding threads, formatting on the host threads, a batch read ahead"}


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(job, code, quality, flag, target_seconds=12.0):
    """The CPU oracle (a port of the reference algorithm, oracle/pheniqs_oracle.c) timed on this host's cores:
    one private decoder set per thread, like the reference's TranscodingThreads (transcode.cpp:1769-1775).
    Every thread decodes 64 Ki-read chunks of the pool, starting at its own offset, until the deadline."""
    import threading
    from oracle import pyoracle
    cores = usable_cores()
    n = code.shape[0]
    chunk = min(65536, n)
    engine = pyoracle.OracleJob(job)
    probe = min(20000, n)
    t0 = time.perf_counter()
    engine.classify(code[:probe], quality[:probe], flag[:probe])
    single = probe / (time.perf_counter() - t0)
    engines = [pyoracle.OracleJob(job) for _ in range(cores)]
    done = [0] * cores
    start = threading.Barrier(cores + 1)

    def work(k):
        start.wait()
        deadline = time.perf_counter() + target_seconds
        position = (k * chunk) % max(1, n - chunk + 1)
        while time.perf_counter() < deadline:
            engines[k].classify(code[position:position + chunk], quality[position:position + chunk], flag[position:position + chunk])
            done[k] += chunk
            position = (position + chunk) % max(1, n - chunk + 1)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in threads:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    elapsed = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / elapsed / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "port",
            "sample": "%d reads (64 Ki-read chunks of the same C2 pool) on %d threads with private decoder sets, %.1f s wall; "
                      "1 thread alone: %.3f Mreads/s" % (total, cores, elapsed, single / 1e6)}


def parity_check(job, code, quality, flag, got, reads):
    """The first `reads` reads of the batch through the oracle: the GPU records must be the same."""
    from oracle import pyoracle
    cores = min(usable_cores(), 64)
    bounds = np.linspace(0, reads, cores + 1).astype(np.int64)
    engines = [pyoracle.OracleJob(job) for _ in range(cores)]
    with ThreadPoolExecutor(cores) as pool:
        results = list(pool.map(lambda k: engines[k].classify(code[bounds[k]:bounds[k + 1]], quality[bounds[k]:bounds[k + 1]],
                                                             flag[bounds[k]:bounds[k + 1]])[0], range(cores)))
    expected = np.concatenate(results)[:, 0]
    for name in ("index", "distance", "hq_distance", "status", "qcfail"):
        assert np.array_equal(got[:reads][name], expected[name]), name
    assert np.abs(got[:reads]["confidence"] - expected["confidence"]).max() < 1e-6
    return reads


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmup", type=int, default=3)
    parser.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU per step")
    parser.add_argument("--pool", type=int, default=1 << 20, help="distinct synthetic reads tiled to --reads")
    parser.add_argument("--algorithm", default="pamld", choices=["pamld", "mdd"])
    parser.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5", "U96", "U384", "V384", "S384", "S192"],
                        help="BASELINE.json config shape; C2 is the headline workload, the others are for exploration (U96 / U384: unique dual index pairs, not in BASELINE.json)")
    parser.add_argument("--priors", default="uniform", choices=["uniform", "estimated"],
                        help="estimated: barcode priors that differ (what a prior estimation pass leaves), not the BASELINE shape")
    parser.add_argument("--fastq", type=int, default=0, metavar="N",
                        help="also time N synthetic reads end to end from FASTQ text to SAM text (next rows f1 + f2)")
    parser.add_argument("--mode", default="fast", choices=["literal", "fast"],
                        help="fast: product form kernel + literal kernel on the ambiguous reads; literal: literal kernel on every read")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--host-staged", type=int, default=0, metavar="READS",
                        help="also time phq_classify_host (pageable host planes -> pinned -> device -> back) on READS reads")
    args = parser.parse_args()

    import torch
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, distributed, synthetic

    rank, local_rank, world = distributed.initialise("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    job, recipe = synthetic.config_job(args.config, args.algorithm)
    if args.priors == "estimated":
        from pheniqs_amd import jobspec
        spec = synthetic.config_spec(args.config, args.algorithm)
        spread = np.random.default_rng(3)
        for _, decoder in jobspec.iter_decoders(spec):
            for record in (decoder.get("codec") or {}).values():
                record["concentration"] = float(10 ** spread.uniform(-0.5, 0.5))
        job = capi.compile_job(spec)
    bytes_per_read = synthetic.algorithmic_bytes_per_read(job)          # 37 for C2
    pool_n = min(args.pool, args.reads)
    tiles = -(-args.reads // pool_n)
    n = args.reads
    code, quality, flag, offsets = synthetic.generate_rows(job, pool_n, recipe["seed"] + 1000 * rank)

    s = DecoderSet(job, device=local_rank)
    s.set_mode(capi.MODE_FAST if args.mode == "fast" else capi.MODE_LITERAL)
    def resident_planes(rows):
        """decoder major planes of the pool, every plane tiled to n reads, concatenated: the layout phq_classify_device takes"""
        pool_planes = torch.from_numpy(s.planes_from_rows(rows, offsets)).to(device)
        parts = []
        for d in s.decoders:
            stride, prefix = d["plane_stride"], d["plane_prefix"]
            if stride:
                parts.append(pool_planes[pool_n * prefix: pool_n * (prefix + stride)].view(pool_n, stride).repeat(tiles, 1)[:n].contiguous().view(-1))
        return torch.cat(parts) if len(parts) > 1 else parts[0]

    base = resident_planes(code)
    qual = resident_planes(quality)
    flags = torch.from_numpy(flag).to(device).repeat(tiles)[:n].contiguous()
    result = torch.zeros(max(1, s.n_result) * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device)
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
    s.profile_enable(True)          # HIP events around every kernel launch, on the launch stream
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(pairs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = s.profile_collect()
    s.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    classify_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))        # all kernels of one step
    dominant = max(kernels, key=lambda name: kernels[name]["ms"])
    kernel_ms = kernels[dominant]["ms"] / kernels[dominant]["launches"]          # the dominant kernel alone

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
            "config": {"workload": "%s: %d reads per GPU, %s, %s, inputs resident in HBM" % (
                           args.config, n, {"C2": "single 8 bp index, 96 barcodes", "C3": "dual 10+10 bp, 96x96 barcodes",
                                            "C4": "sci-RNA-seq cellular [96x10, 96x(10+10)] + UMI 8", "C5": "SPLiT-seq cellular [96x8 x3, 4x6 rc] + UMI 10",
                                            "U96": "unique dual index, 96 pairs of 8+8 bp", "U384": "unique dual index, 384 pairs of 8+8 bp",
                                            "V384": "unique dual index, 384 pairs of 10+10 bp", "S384": "single 8 bp index, 384 barcodes", "S192": "single 8 bp index, 192 barcodes"}[args.config],
                           args.algorithm.upper()),
                       "reads_per_gpu": n, "pool": pool_n, "kernel_mode": args.mode,
                       "sharding": "reads x%d, counters merged by one all-reduce per step" % world},
            "roofline": {"bound": "hbm", "achieved": bytes_per_read * n / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bytes_per_read * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": profiled_traffic(dominant, n, args.config, args.algorithm, args.mode),
                         "kernel": dominant, "kernel_ms": kernel_ms, "algorithmic_bytes_per_read": bytes_per_read,
                         "measured_copy_GBs": measured_copy_bandwidth(torch, device) if world == 1 else None,
                         "all_kernels_ms_per_step": {name: v["ms"] / args.steps for name, v in kernels.items()},
                         "classify_ms_per_step": classify_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(job, code, quality, flag)
            got = result[: pool_n * RESULT_DTYPE.itemsize].cpu().numpy().view(RESULT_DTYPE)      # first result plane, first pool pass
            line["parity_checked_reads"] = parity_check(job, code, quality, flag, got, min(pool_n, 1 << 19))
        if args.host_staged:
            # PCIe inclusive rate; never the headline value
            m = args.host_staged
            reps = -(-m // pool_n)
            host_base = s.planes_from_rows(np.tile(code, (reps, 1))[:m], offsets)
            host_qual = s.planes_from_rows(np.tile(quality, (reps, 1))[:m], offsets)
            host_flag = np.tile(flag, reps)[:m].copy()
            s.classify_host(m, host_base, host_qual, host_flag)            # allocates the pinned slots
            t0 = time.perf_counter()
            s.classify_host(m, host_base, host_qual, host_flag)
            dt = time.perf_counter() - t0
            line["host_staged"] = {"reads": m, "Mreads/s": m / dt / 1e6, "GB/s_over_pcie": m * bytes_per_read / dt / 1e9,
                                   "note": "pageable host planes, memcpy to pinned slots, 2 streams, results copied back"}
            # the same batch in pinned memory end to end (phq_host_alloc): no bounce copy on either side
            pinned = {}
            for name, array in (("base", host_base), ("qual", host_qual), ("flag", host_flag)):
                pinned[name] = torch.from_numpy(array).pin_memory()
            out_result = torch.zeros(max(1, s.n_result) * m * RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
            out_word = torch.zeros(m, dtype=torch.int32).pin_memory()
            call = lambda: s._check(s.lib.phq_classify_host(s.handle, m, m, pinned["base"].data_ptr(), pinned["qual"].data_ptr(),
                                                            pinned["flag"].data_ptr(), out_result.data_ptr(), out_word.data_ptr()))
            call()
            t0 = time.perf_counter()
            call()
            dt = time.perf_counter() - t0
            line["host_staged_pinned"] = {"reads": m, "Mreads/s": m / dt / 1e6, "GB/s_over_pcie": m * bytes_per_read / dt / 1e9,
                                          "note": "caller's planes and result buffers pinned: async copies straight from / to them"}
        if args.fastq:
            line["fastq_to_sam"] = fastq_end_to_end(args, code, quality, flag, args.fastq)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
