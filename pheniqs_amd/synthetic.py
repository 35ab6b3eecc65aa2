"""Seeded synthetic workloads of the shapes BASELINE.json names.

Barcode sets: distinct random ACGT words with pairwise Hamming distance >= 3, sorted (sorted codec keys
give index 1..N, reference transcode.cpp:988-1004).  Reads: the true barcode is drawn from the priors,
or with probability `noise` a uniform random word; per base quality is an Illumina-like mix
{Q37 80 %, Q25 10 %, Q12 7 %, Q2 3 %}; a base is substituted with probability 10^(-q/10); 0.1 % of
bases read N at Q2; 3 % of reads carry an incoming QC-fail flag.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import json
import os

import numpy as np

from . import capi

TOPICS = ("sample", "molecular", "cellular")


def iter_decoders(job: Dict[str, Any]):
    """(kind, decoder) in the order TranscodingDecoder::classify runs them: transcode.h:51-60."""
    for kind in TOPICS:
        node = job.get(kind)
        if node is None:
            continue
        if isinstance(node, dict):
            yield kind, node
        else:
            for element in node:
                yield kind, element


BASE_SEED = 20240613
ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
BAM = np.array([1, 2, 4, 8], dtype=np.uint8)
QUALITY_VALUES = np.array([37, 25, 12, 2], dtype=np.uint8)
QUALITY_WEIGHTS = np.array([0.80, 0.10, 0.07, 0.03])


def barcode_set(rng: np.random.Generator, count: int, length: int, min_distance: int = 3) -> List[str]:
    """Rejection sampling of `count` words of `length` nucleotides with pairwise Hamming >= min_distance."""
    if 4 ** length < count:
        raise ValueError("cannot draw %d distinct words of length %d" % (count, length))
    chosen = np.zeros((0, length), dtype=np.uint8)
    while chosen.shape[0] < count:
        candidate = rng.integers(0, 4, size=(1, length), dtype=np.uint8)
        if chosen.shape[0] == 0 or (chosen != candidate).sum(axis=1).min() >= min_distance:
            chosen = np.concatenate([chosen, candidate])
    words = sorted("".join("ACGT"[c] for c in row) for row in chosen)
    return words


def decoder_spec(kind_algorithm: str, segments: Sequence[Sequence[str]], tokens: Sequence[str], knit: Optional[Sequence[str]] = None,
                 pairing: str = "product", **extra) -> Dict[str, Any]:
    """User level decoder spec.  `segments[s]` is the word list of barcode segment s; `product` builds the
    full combinatorial codec (96 x 96 = 9216 for the dual index shape), `zip` pairs word i with word i."""
    if pairing == "product":
        combos = [[w] for w in segments[0]]
        for words in segments[1:]:
            combos = [c + [w] for c in combos for w in words]
    else:
        combos = [list(c) for c in zip(*segments)]
    codec = {"@" + "".join(c): {"barcode": c, "concentration": 1} for c in combos}
    spec = {"algorithm": kind_algorithm, "transform": {"token": list(tokens)}, "codec": codec}
    if knit:
        spec["transform"]["knit"] = list(knit)
    spec.update(extra)
    return spec


# BASELINE.json's C2..C5 (C4V: C4 with the 196 pairs of the reference's sci-RNA-seq vignette) plus shapes outside it that real runs use a lot: unique dual indexes, 96 or 384 PAIRS of
# 8 + 8 nt in one two segment sample decoder (not every combination: pair i is word i of each list)
CONFIG_NUMBER = {"C2": 2, "C3": 3, "C4": 4, "C5": 5, "U96": 6, "U384": 7, "V384": 8, "S384": 9, "S192": 10, "C4V": 11, "H94": 12, "H94A": 13, "L12": 14, "L16": 15}
# H94 / H94A: the one workload the reference publishes a wall clock for (docs/illumina_api_vignette.md:222, lane 1 of flowcell
# H7LT2DSXX): 94 unique dual index pairs of 8 + 8 nt, PAMLD — H94 as its prior estimation run has them (uniform priors, noise
# 0.05), H94A with the priors and the noise that run left (example/H7LT2DSXX/H7LT2DSXX_l01_adjusted.json).  The barcodes and
# priors are the reference's own data, kept as a fixture (tests/tools/make_h7lt2dsxx_fixture.py writes it).
H7LT2DSXX_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "H7LT2DSXX", "l01_sample_decoder.json")


def config_job(name: str, algorithm: str = "pamld") -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """Compiled job + generation recipe for BASELINE config C2..C5.  seed = 20240613 + config number."""
    number = CONFIG_NUMBER[name]
    # compiled by the library's own configuration compiler (phq_compile_job); tests/jobspec.py is its independent Python
    # restatement, which the tests hold against it
    return capi.compile_job(config_spec(name, algorithm)), {"name": name, "seed": BASE_SEED + number}


def config_spec(name: str, algorithm: str = "pamld") -> Dict[str, Any]:
    """The user level (uncompiled) job document of BASELINE config C2..C5."""
    number = CONFIG_NUMBER[name]
    rng = np.random.default_rng(BASE_SEED + number)
    if name == "C2":        # single 8 bp i7 index, 96 barcodes
        words = barcode_set(rng, 96, 8)
        spec = {"input segment cardinality": 2,
                "sample": decoder_spec(algorithm, [words], ["1::8"])}
    elif name == "C3":      # dual 10 bp i7 + i5, 96 x 96 combinatorial, one sample decoder with two segments
        i7, i5 = barcode_set(rng, 96, 10), barcode_set(rng, 96, 10)
        spec = {"input segment cardinality": 3,
                "sample": decoder_spec(algorithm, [i7, i5], ["1::10", "2::10"])}
    elif name in ("C4", "C4V"):      # sci-RNA-seq: cellular [96 x 10, 96 pairs (10 + 10)] + naive UMI 8; C4V: the vignette's 196 pairs
        pairs = 96 if name == "C4" else 196
        a, b, c = barcode_set(rng, 96, 10), barcode_set(rng, pairs, 10), barcode_set(rng, pairs, 10)
        spec = {"input segment cardinality": 3,
                "molecular": [{"algorithm": "naive", "transform": {"token": ["0::8"]}}],
                "cellular": [decoder_spec(algorithm, [a], ["0:8:18"]),
                             decoder_spec(algorithm, [b, c], ["1::10", "2::10"], pairing="zip")]}
    elif name in ("U96", "U384"):      # unique dual index pairs, i7 + i5 of 8 nt each
        pairs = 96 if name == "U96" else 384
        i7, i5 = barcode_set(rng, pairs, 8), barcode_set(rng, pairs, 8)
        spec = {"input segment cardinality": 3,
                "sample": decoder_spec(algorithm, [i7, i5], ["1::8", "2::8"], pairing="zip")}
    elif name == "V384":               # 384 unique dual index pairs of 10 + 10 nt
        i7, i5 = barcode_set(rng, 384, 10), barcode_set(rng, 384, 10)
        spec = {"input segment cardinality": 3,
                "sample": decoder_spec(algorithm, [i7, i5], ["1::10", "2::10"], pairing="zip")}
    elif name in ("H94", "H94A"):
        fixture = json.load(open(H7LT2DSXX_FIXTURE))
        which = fixture["estimate" if name == "H94" else "adjusted"]
        codec = {key: {"barcode": list(record["barcode"]), "concentration": 1 if name == "H94" else record["concentration"]}
                 for key, record in fixture["codec"].items()}
        spec = {"input segment cardinality": 3,
                "sample": {"algorithm": algorithm, "noise": which["noise"], "confidence threshold": which["confidence threshold"],
                           "transform": {"token": ["1::8", "2::8"]}, "codec": codec}}
    elif name in ("L12", "L16"):       # 96 single indexes of 12 / 16 nt: past the streaming kernel's 10 nt (the paired tier in two parts)
        spec = {"input segment cardinality": 2,
                "sample": decoder_spec(algorithm, [barcode_set(rng, 96, 12 if name == "L12" else 16, min_distance=4)], ["1::%d" % (12 if name == "L12" else 16)])}
    elif name in ("S384", "S192"):     # 384 / 192 single 8 nt indexes
        spec = {"input segment cardinality": 2,
                "sample": decoder_spec(algorithm, [barcode_set(rng, 384 if name == "S384" else 192, 8)], ["1::8"])}
    else:                   # C5 SPLiT-seq: cellular [96 x 8] x 3 + 4 x 6 reverse complemented, naive UMI 10
        r1, r2, r3 = barcode_set(rng, 96, 8), barcode_set(rng, 96, 8), barcode_set(rng, 96, 8)
        r4 = barcode_set(rng, 4, 6)
        spec = {"input segment cardinality": 2,
                "molecular": [{"algorithm": "naive", "transform": {"token": ["1::10"]}}],
                "cellular": [decoder_spec(algorithm, [r1], ["1:10:18"]),
                             decoder_spec(algorithm, [r2], ["1:48:56"]),
                             decoder_spec(algorithm, [r3], ["1:86:94"]),
                             decoder_spec(algorithm, [r4], ["0::6"], knit=["~0"])]}
    return spec


def _codec_arrays(decoder: Dict[str, Any]) -> Tuple[np.ndarray, np.ndarray]:
    records = list((decoder.get("codec") or {}).values())
    L = int(decoder["nucleotide cardinality"])
    if not records:
        return np.zeros((0, L), dtype=np.uint8), np.zeros(0)
    text = np.frombuffer("".join("".join(r["barcode"]) for r in records).encode(), dtype=np.uint8).reshape(len(records), L)
    index = np.zeros(256, dtype=np.uint8)
    index[ASCII] = np.arange(4, dtype=np.uint8)
    return index[text], np.array([float(r["concentration"]) for r in records])


def generate_rows(job: Dict[str, Any], n: int, seed: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[int]]:
    """n reads as row major staged observations: (code[n, sum L], quality[n, sum L], flag[n], decoder offsets).
    Each decoder's observation is generated independently (what Rule::apply would have cut out of the read)."""
    rng = np.random.default_rng(seed)
    columns_code, columns_quality, offsets = [], [], []
    offset = 0
    for _, decoder in iter_decoders(job):
        L = int(decoder["nucleotide cardinality"])
        offsets.append(offset)
        offset += L
        words, prior = _codec_arrays(decoder)
        noise = float(decoder.get("noise", 0.01))
        truth = rng.integers(0, 4, size=(n, L), dtype=np.uint8)            # noise reads and codec-less (UMI) decoders
        if words.shape[0] > 0:
            pick = rng.choice(words.shape[0], size=n, p=prior / prior.sum())
            from_codec = rng.random(n) >= noise
            truth[from_codec] = words[pick[from_codec]]
        quality = QUALITY_VALUES[rng.choice(4, size=(n, L), p=QUALITY_WEIGHTS)]
        error = rng.random((n, L)) < np.power(10.0, -quality.astype(np.float64) / 10.0)
        shift = rng.integers(1, 4, size=(n, L), dtype=np.uint8)
        called = np.where(error, (truth + shift) & 3, truth)
        code = BAM[called]
        unknown = rng.random((n, L)) < 0.001
        code[unknown] = 15
        quality = quality.copy()
        quality[unknown] = 2
        columns_code.append(code)
        columns_quality.append(quality)
    flag = (rng.random(n) < 0.03).astype(np.uint8)
    if columns_code:
        return np.concatenate(columns_code, axis=1), np.concatenate(columns_quality, axis=1), flag, offsets
    return np.zeros((n, 0), dtype=np.uint8), np.zeros((n, 0), dtype=np.uint8), flag, offsets


def algorithmic_bytes_per_read(job: Dict[str, Any]) -> int:
    """2 * L_total (base + quality) + 1 (input flag) + 16 per PAMLD/MDD decoder + 4 (status word)."""
    total = 0
    closed = 0
    for _, decoder in iter_decoders(job):
        total += int(decoder["nucleotide cardinality"])
        if decoder["algorithm"] in ("pamld", "mdd"):
            closed += 1
    return 2 * total + 1 + 16 * closed + 4


def generate_planes_device(torch, job: Dict[str, Any], decoders: Sequence[Dict[str, int]], plane_width: int, n: int, seed: int, device,
                           host_rows: int = 1 << 20, chunk: int = 1 << 23):
    """The recipe of generate_rows drawn ON THE DEVICE, straight into the decoder major planes phq_classify_device takes:
    every one of the n reads is its own draw (no pool tiled to size), so nothing in a measurement can ride on repeated
    rows.  Same distributions as generate_rows, another random stream (torch's).  Returns (base, quality, flag) device
    tensors and the first `host_rows` reads back as generate_rows' row major host arrays (code, quality, flag, offsets)
    for the CPU baseline and the parity check.  `decoders` = DecoderSet.decoders, `plane_width` = DecoderSet.plane_width."""
    base = torch.zeros(max(1, n * max(plane_width, 1)), dtype=torch.uint8, device=device)
    qual = torch.zeros_like(base)
    flag = torch.empty(n, dtype=torch.uint8, device=device)
    bam = torch.tensor(BAM, dtype=torch.uint8, device=device)
    values = torch.tensor(QUALITY_VALUES, dtype=torch.uint8, device=device)
    error_rate = torch.tensor(np.power(10.0, -QUALITY_VALUES.astype(np.float64) / 10.0), dtype=torch.float32, device=device)
    edges = torch.tensor(np.cumsum(QUALITY_WEIGHTS)[:-1], dtype=torch.float32, device=device)
    specs = []
    offsets: List[int] = []
    offset = 0
    for (_, decoder), info in zip(iter_decoders(job), decoders):
        L = int(decoder["nucleotide cardinality"])
        offsets.append(offset)
        offset += L
        words, prior = _codec_arrays(decoder)
        specs.append((L, info, torch.from_numpy(words).to(device) if words.shape[0] else None,
                      torch.tensor(np.cumsum(prior / prior.sum()), dtype=torch.float64, device=device) if words.shape[0] else None,
                      float(decoder.get("noise", 0.01))))
    generator = torch.Generator(device=device)
    for number, first in enumerate(range(0, n, chunk)):
        m = min(chunk, n - first)
        generator.manual_seed(seed * 1_000_003 + number)
        for L, info, words, cumulative, noise in specs:
            if L == 0 or info["plane_stride"] == 0:
                continue
            truth = torch.randint(0, 4, (m, L), generator=generator, device=device, dtype=torch.uint8)
            if words is not None:
                pick = torch.searchsorted(cumulative, torch.rand(m, generator=generator, device=device, dtype=torch.float64)).clamp_(max=words.shape[0] - 1)
                from_codec = torch.rand(m, generator=generator, device=device) >= noise
                truth = torch.where(from_codec[:, None], words[pick], truth)
                del pick, from_codec
            klass = torch.bucketize(torch.rand(m, L, generator=generator, device=device), edges, right=True)
            error = torch.rand(m, L, generator=generator, device=device) < error_rate[klass]
            shift = torch.randint(1, 4, (m, L), generator=generator, device=device, dtype=torch.uint8)
            called = torch.where(error, (truth + shift) & 3, truth)
            code = bam[called.long()]
            quality = values[klass]
            unknown = torch.rand(m, L, generator=generator, device=device) < 0.001
            code[unknown] = 15
            quality[unknown] = 2
            stride, prefix = info["plane_stride"], info["plane_prefix"]
            base[n * prefix: n * (prefix + stride)].view(n, stride)[first:first + m, :L] = code
            qual[n * prefix: n * (prefix + stride)].view(n, stride)[first:first + m, :L] = quality
            del truth, klass, error, shift, called, code, quality, unknown
        flag[first:first + m] = (torch.rand(m, generator=generator, device=device) < 0.03).to(torch.uint8)
    rows = min(host_rows, n)
    code_rows = np.zeros((rows, offset), dtype=np.uint8)
    quality_rows = np.zeros((rows, offset), dtype=np.uint8)
    for (L, info, _, _, _), at in zip(specs, offsets):
        if L == 0 or info["plane_stride"] == 0:
            continue
        stride, prefix = info["plane_stride"], info["plane_prefix"]
        code_rows[:, at:at + L] = base[n * prefix: n * (prefix + stride)].view(n, stride)[:rows, :L].cpu().numpy()
        quality_rows[:, at:at + L] = qual[n * prefix: n * (prefix + stride)].view(n, stride)[:rows, :L].cpu().numpy()
    return base, qual, flag, (code_rows, quality_rows, flag[:rows].cpu().numpy(), offsets)


def rows_from_planes(torch, job: Dict[str, Any], decoders: Sequence[Dict[str, int]], n: int, base, qual, flag, ordinals: np.ndarray):
    """The reads `ordinals` of a batch resident in decoder major planes (n reads a plane) back as generate_rows' row major host
    arrays (code, quality, flag): what a parity check hands the oracle when the sample is spread over the whole batch."""
    index = torch.from_numpy(np.ascontiguousarray(ordinals, dtype=np.int64)).to(base.device)
    width = sum(int(decoder["nucleotide cardinality"]) for _, decoder in iter_decoders(job))
    code_rows = np.zeros((index.numel(), width), dtype=np.uint8)
    quality_rows = np.zeros((index.numel(), width), dtype=np.uint8)
    at = 0
    for (_, decoder), info in zip(iter_decoders(job), decoders):
        L = int(decoder["nucleotide cardinality"])
        stride, prefix = info["plane_stride"], info["plane_prefix"]
        if L and stride:
            code_rows[:, at:at + L] = base[n * prefix: n * (prefix + stride)].view(n, stride)[index][:, :L].cpu().numpy()
            quality_rows[:, at:at + L] = qual[n * prefix: n * (prefix + stride)].view(n, stride)[index][:, :L].cpu().numpy()
        at += L
    flags = flag[index].cpu().numpy() if flag is not None else np.zeros(index.numel(), dtype=np.uint8)
    return code_rows, quality_rows, flags
