#!/usr/bin/env python3
"""The table shape -> kernel chain -> rate over a GRID of decoder shapes instead of a handful of named ones: barcode length L x
barcodes N x {single index, unique pairs, two and three segment combinatorial sets} x {uniform, estimated priors}, PAMLD, fast mode.
Every kernel name is the dispatcher's own (phq_profile_chain), so a `*_supports()` predicate boundary that hides a 4x step shows
as a row whose first kernel is not a tier 0 one.

usage (GPU box, repo root):  python profiles/tools/shape_grid.py [--reads 16000000] [--out profiles/rNN_shape_grid.md] [--batches]

`--batches` adds the small batch table: C2 .. C5 at 64 Ki, 1 Mi, 4 Mi and 16 Mi reads a call.
tests/test_gpu_shape_grid.py walks the same grid (shapes()) at 64 Ki reads and holds every shape to a tier 0 kernel or to its
place on the allow list below."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

LENGTHS = (6, 8, 10, 12, 16, 20)
COUNTS = (8, 24, 96, 192, 288, 384, 768, 1536)
FAMILIES = ("single", "paired", "combinatorial2", "combinatorial3")
PRIORS = ("uniform", "estimated")
# kernels that finish most reads from a lookup and a short list (or, for sets of a few words, score every word once): tier 0.
# Anything else as the FIRST kernel of a chain is the product form over all barcodes or the literal kernel for every read.
TIER0 = ("pamld_stream_kernel", "pamld_cross_stream_kernel", "pamld_pair_kernel", "pamld_plate_kernel", "pamld_near_kernel",
         "pamld_cross_near_kernel", "pamld_few_kernel", "pamld_tiny_kernel")
# Shapes WITHOUT a tier 0, by (family, predicate) with the reason: the test fails on a shape that has none and is not here.
def crowded(L, words):
    """the grid draws its words at least three mismatches apart where 4^L has room for that, two or one where it has not"""
    return words * 256 > 4 ** L


ALLOW = (
    ("single", lambda L, N, priors: L <= 10 and N > 1024 and priors == "estimated", "more than some 800 barcodes of up to 10 nt with priors that differ: the list kernel's records with their prior ratios do not fit a CU, not even in its 512 lane layout"),
    ("single", lambda L, N, priors: crowded(L, N), "words the grid could not draw three mismatches apart (4^L too small): no list or distance bound certifies anything, the product form is the first kernel"),
    ("paired", lambda L, N, priors: 2 * L > 32, "pairs of 20 + 20 nt: 40 positions are past the 32 every tier stops at; the general literal kernel (pamld_literal_long.hip) walks every barcode for every read"),
    ("paired", lambda L, N, priors: L == 6 and N >= 96, "pairs of 6 + 6 nt: 96 and more sit closer than the six mismatches the paired tier's bound needs for half of them"),
    ("paired", lambda L, N, priors: crowded(L, N), "index words the grid could not draw three mismatches apart: pairs closer than six mismatches"),
    ("combinatorial2", lambda L, N, priors: 2 * L > 32, "two segments of 20 nt: 40 positions, the general literal kernel for every read"),
    ("combinatorial2", lambda L, N, priors: L > 10, "two segment sets of 12 and 16 nt words: the streaming kernel of sets tables words of 8 to 10 nt; the factorised product form is the first kernel"),
    ("combinatorial2", lambda L, N, priors: L == 6 and N > 256, "two segment sets of 6 nt words past a plate: under the streaming kernel of sets' 8 nt"),
    ("combinatorial2", lambda L, N, priors: L == 10 and N <= 256, "up to 16 words of 10 nt a segment: plates are 4 to 8 nt; the factorised product form (15-19 Greads/s) is the first kernel"),
    ("combinatorial3", lambda L, N, priors: True, "three segment sets: no tier 0 yet (DESIGN.md §7); the factorised product form (8-17 Greads/s) with uniform priors, the two word one with priors that differ; past 32 nt (3 x 12 and more) the general literal kernel for every read"),
)


def feasible(family, L, N):
    """segment word counts of the shape, or None where the words cannot exist (4^L too small for the distance asked)"""
    if family == "single" or family == "paired":
        per = [N] * (1 if family == "single" else 2)
    elif family == "combinatorial2":
        side = int(np.ceil(np.sqrt(N)))
        per = [side, side]
    else:
        side = int(np.ceil(N ** (1.0 / 3.0)))
        per = [side, side, side]
    for words in per:
        if words * 16 > 4 ** L:            # a sixteenth of the space at most: the rejection sampling ends
            return None
    return per


def shapes():
    for family in FAMILIES:
        for L in LENGTHS:
            for N in COUNTS:
                per = feasible(family, L, N)
                if per is not None:
                    for priors in PRIORS:
                        yield family, L, N, per, priors


def allowed(family, L, N, priors="uniform"):
    words = max(feasible(family, L, N) or [N])
    for which, predicate, reason in ALLOW:
        if which == family and predicate(L, words if "draw" in reason else N, priors):
            return reason
    return None


def spec_of(family, L, N, per, priors, seed=11):
    from pheniqs_amd import synthetic
    rng = np.random.default_rng(seed + 131 * L + N)
    distance = 3 if max(per) * 256 <= 4 ** L else 2 if max(per) * 64 <= 4 ** L else 1
    segments = [synthetic.barcode_set(rng, words, L, min_distance=distance) for words in per]
    tokens = ["%d::%d" % (k + 1, L) for k in range(len(per))] if family != "single" else ["1::%d" % L]
    decoder = synthetic.decoder_spec("pamld", segments, tokens, pairing="zip" if family == "paired" else "product")
    if priors == "estimated":
        spread = np.random.default_rng(3)
        for record in decoder["codec"].values():
            record["concentration"] = float(10 ** spread.uniform(-0.5, 0.5))
    return {"input segment cardinality": len(per) + 1, "sample": decoder}, int(np.prod(per)) if family.startswith("comb") else N


def plan(spec, reads=100_000_000):
    """[kernel, ...] of one shape as phq_plan_chain names them: the stage dispatcher with its launches off, no GPU needed"""
    from pheniqs_amd import DecoderSet, capi
    s = DecoderSet(spec, device=capi.DEVICE_NONE)
    try:
        return [c["kernel"] for c in s.plan_chain(reads, capi.MODE_FAST)]
    finally:
        s.close()


def run(torch, spec, reads, steps=3, warmup=1, events=True):
    """(Greads/s, ms a step, [(kernel, ms a step)]) of one shape, inputs resident in HBM, every read its own draw; events = False:
    nothing recorded between the launches (phq_profile_chain's events cost a small batch a third of its time), no chain"""
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic
    job = capi.compile_job(spec)
    s = DecoderSet(spec, device=0)
    s.set_mode(capi.MODE_FAST)
    base, qual, flag, _ = synthetic.generate_planes_device(torch, job, s.decoders, s.plane_width, reads, 3, "cuda", host_rows=1)
    result = torch.zeros(max(1, s.n_result) * reads * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    word = torch.zeros(reads, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    s.profile_enable(events)
    for _ in range(warmup):
        s.classify_device(reads, 0, reads, base, qual, flag, result, word, stream)
    torch.cuda.synchronize()
    if events:
        s.profile_chain()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.classify_device(reads, 0, reads, base, qual, flag, result, word, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    chain, order = {}, []
    spans = s.profile_chain() if events else []
    # what phq_plan_chain says the dispatcher WOULD do is what it did (one decision, read twice)
    planned = [(c["decoder"], c["kernel"]) for c in s.plan_chain(reads, capi.MODE_FAST)]
    assert not events or planned == [(c["decoder"], c["kernel"]) for c in spans[:len(spans) // steps]], (planned, spans)
    for span in spans:
        if span["kernel"] not in chain:
            chain[span["kernel"]] = 0.0
            order.append(span["kernel"])
        chain[span["kernel"]] += span["ms"] / steps
    bytes_per_read = synthetic.algorithmic_bytes_per_read(job)
    s.close()
    del base, qual, flag, result, word
    torch.cuda.empty_cache()
    return reads / dt / 1e9, dt * 1e3, [(k, chain[k]) for k in order], bytes_per_read


def batches(torch, out):
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic
    out.append("\n## Small batches: Greads/s by reads a call, calls in a row on one stream, no events between the launches (the reference hands reads over 2 048 at a time, configuration.json:369; by kernel and as a graph: `small_batch.py`)\n")
    out.append("| shape | 64 Ki | 1 Mi | 4 Mi | 16 Mi |\n|---|---|---|---|---|")
    for config in ("C2", "C3", "C4", "C5"):
        cells = []
        for reads in (1 << 16, 1 << 20, 1 << 22, 1 << 24):
            spec = synthetic.config_spec(config, "pamld")
            rate, ms, chain, _ = run(torch, spec, reads, steps=200 if reads <= (1 << 20) else 20, warmup=3, events=False)
            cells.append("%.2f (%.3f ms)" % (rate, ms))
        out.append("| %s | %s |" % (config, " | ".join(cells)))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--reads", type=int, default=16_000_000)
    parser.add_argument("--out", default=None)
    parser.add_argument("--batches", action="store_true")
    parser.add_argument("--only", default=None, help="family name: that family's shapes alone")
    args = parser.parse_args()
    import torch
    from pheniqs_amd import capi
    lines = ["# Shape grid -> kernel chain -> rate (build %s)\n" % capi.build_id()[:16],
             "Generated by `profiles/tools/shape_grid.py --reads %d`: PAMLD, fast mode, one MI355X, inputs resident in HBM, every read its own draw; kernel names and" % args.reads,
             "times are `phq_profile_chain`'s.  `of roofline` = algorithmic bytes (SURVEY.md §8d) over the step's time against 8 TB/s.  **bold** first kernel = no tier 0",
             "(the product form over all barcodes, or the literal kernel, sees every read); such a shape is on `ALLOW` in the tool with its reason, or the GPU suite fails.\n",
             "| family | L | barcodes | priors | Greads/s | ms / step | of roofline | kernels of a step (ms) |", "|---|---|---|---|---|---|---|---|"]
    worst = []
    for family, L, N, per, priors in shapes():
        if args.only and family != args.only:
            continue
        spec, barcodes = spec_of(family, L, N, per, priors)
        try:
            rate, ms, chain, bytes_per_read = run(torch, spec, args.reads)
        except Exception as error:                   # a shape the library refuses: say so in the table
            lines.append("| %s | %d | %d | %s | - | - | - | refused: %s |" % (family, L, barcodes, priors, str(error)[:80]))
            continue
        first = chain[0][0] if chain else "-"
        tier0 = any(first.startswith(name) for name in TIER0)
        text = " -> ".join("%s %.3f" % (("**%s**" % k) if (i == 0 and not tier0) else k, t) for i, (k, t) in enumerate(chain))
        frac = bytes_per_read * args.reads / (ms * 1e-3) / 8e12
        lines.append("| %s | %d | %d | %s | %.1f | %.3f | %.3f | %s |" % (family, L, barcodes, priors, rate, ms, frac, text.replace("pamld_", "").replace("_kernel", "")))
        worst.append((frac, family, L, barcodes, priors, first))
        print(lines[-1], flush=True)
    worst.sort()
    lines.append("\nThe ten shapes furthest below the roofline: " + "; ".join("%s L=%d N=%d %s %.3f (%s)" % (f, L, N, p, frac, k.replace("pamld_", "").replace("_kernel", "")) for frac, f, L, N, p, k in worst[:10]))
    if args.batches:
        batches(torch, lines)
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
