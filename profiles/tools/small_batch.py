"""Small batches: what a call of N reads costs when calls follow one another on one stream — launched one by one, and as a
captured graph (hipGraph through torch.cuda.CUDAGraph: the whole chain of a call replayed as one submission).

The shape grid's small batch table is taken with phq_profile_chain's events around every kernel; this one has nothing between
the launches, so the difference between the two columns is what a graph would save and the rest is the kernels' own time.

usage (GPU box, repo root):  python profiles/tools/small_batch.py [--configs C2,C3] [--out gpurun_out/rNN_small_batch.md]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def measure(torch, config, reads, steps):
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic
    spec = synthetic.config_spec(config, "pamld")
    job = capi.compile_job(spec)
    s = DecoderSet(spec, device=0)
    s.set_mode(capi.MODE_FAST)
    base, qual, flag, _ = synthetic.generate_planes_device(torch, job, s.decoders, s.plane_width, reads, 3, "cuda", host_rows=1)
    result = torch.zeros(max(1, s.n_result) * reads * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    word = torch.zeros(reads, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        handle = side.cuda_stream
        for _ in range(5):
            s.classify_device(reads, 0, reads, base, qual, flag, result, word, handle)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.classify_device(reads, 0, reads, base, qual, flag, result, word, handle)
        side.synchronize()
        direct = (time.perf_counter() - t0) / steps
        # one call, alone: launch to finished
        alone = []
        for _ in range(20):
            t0 = time.perf_counter()
            s.classify_device(reads, 0, reads, base, qual, flag, result, word, handle)
            side.synchronize()
            alone.append(time.perf_counter() - t0)
        alone.sort()
    replay = None
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            s.classify_device(reads, 0, reads, base, qual, flag, result, word, torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / steps
        del graph
    except Exception as error:                                 # a capture the runtime refuses: say so
        print("capture refused for %s at %d reads: %s" % (config, reads, str(error)[:200]), file=sys.stderr)
    s.profile_enable(True)
    s.classify_device(reads, 0, reads, base, qual, flag, result, word, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    spans = s.profile_chain()
    s.close()
    return direct, alone[len(alone) // 2], replay, spans


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--configs", default="C2,C3")
    parser.add_argument("--out", default=None)
    args = parser.parse_args()
    import torch
    from pheniqs_amd import capi
    lines = ["# Small batches, launch by launch and as a graph (build %s)\n" % capi.build_id()[:16],
             "PAMLD, fast mode, one MI355X, inputs resident in HBM.  `in a row` = calls following one another on one stream, nothing between the launches;",
             "`alone` = one call from its first launch to the stream's end (median of 20); `graph` = the call captured once and replayed; `kernels` = the chain's own",
             "time by events around each launch.\n",
             "| shape | reads a call | in a row: us, Greads/s | alone: us | graph: us, Greads/s | kernels (us) |", "|---|---|---|---|---|---|"]
    for config in args.configs.split(","):
        for reads in (1 << 11, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
            direct, alone, replay, spans = measure(torch, config, reads, 400 if reads <= (1 << 20) else 100)
            chain = " + ".join("%s %.1f" % (c["kernel"].replace("pamld_", "").replace("_kernel", ""), c["ms"] * 1e3) for c in spans)
            lines.append("| %s | %d | %.1f, %.2f | %.1f | %s | %s |" % (config, reads, direct * 1e6, reads / direct / 1e9, alone * 1e6,
                                                                    ("%.1f, %.2f" % (replay * 1e6, reads / replay / 1e9)) if replay else "refused", chain))
            print(lines[-1], flush=True)
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
