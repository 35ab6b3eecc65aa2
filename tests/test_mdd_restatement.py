"""The C oracle's MDD against a second restatement that shares nothing with it (tests/mdd_restatement.py), 10^5 random
reads per case: quality masking, tolerances at and past the Shannon bound, two and three segment barcodes,
reverse complemented and knitted tokens, short reads, N.  Not a pin — the reference holds no vector for any of these
 — but two restatements by different routes that agree."""
import numpy as np
import pytest

from oracle import pyoracle
from pheniqs_amd import capi, synthetic

import mdd_restatement as second

READS = 100_000
LETTERS = np.array([1, 2, 4, 8], dtype=np.uint8)


def draw_reads(rng, spec_decoder, job_decoder, segment_lengths, n, n_input, read_length):
    """Input reads whose token slices hold a barcode (or a random word), with errors, N and a share of short reads."""
    rule = second.parse_rule(job_decoder["transform"])
    records = list(job_decoder["codec"].values())
    words = ["".join(r["barcode"]) for r in records]
    reads = []
    for r in range(n):
        segments = [(LETTERS[rng.integers(0, 4, read_length)].copy(), rng.choice(np.array([2, 12, 25, 37], dtype=np.uint8), read_length, p=[0.05, 0.15, 0.2, 0.6])) for _ in range(n_input)]
        if rng.random() > 0.05:
            word = words[rng.integers(len(words))]
            at = 0
            # write the barcode where the rule reads it: undo the knit piece by piece
            for k, pieces in enumerate(rule):
                for segment, start, end, flipped in pieces:
                    code, _ = segments[segment]
                    first, last = second.bounds(start, end, len(code))
                    piece = np.array([second.ASCII_TO_BAM[c] for c in word[at:at + last - first]], dtype=np.uint8)
                    if flipped:
                        piece = second.REVERSE_COMPLEMENT[piece[::-1]]
                    code[first:first + len(piece)] = piece
                    at += last - first
        for code, quality in segments:
            wrong = rng.random(len(code)) < 0.04
            code[wrong] = LETTERS[rng.integers(0, 4, int(wrong.sum()))]
            unknown = rng.random(len(code)) < 0.003
            code[unknown] = 15
            quality[unknown] = 2
        if rng.random() < 0.02:                                  # a short read: the observation comes out short
            k = rng.integers(n_input)
            cut = rng.integers(0, read_length)
            segments[k] = (segments[k][0][:cut], segments[k][1][:cut])
        reads.append(segments)
    return rule, reads


CASES = {
    # name: (segment word counts and lengths, tokens, knit, extra, input segments, input read length)
    "single_tolerance_1": ([(24, 8)], ["0:2:10"], None, {"distance tolerance": [1]}, 1, 14),
    "single_past_the_shannon_bound": ([(24, 8)], ["0::8"], None, {"distance tolerance": [3]}, 1, 10),
    "masking_20": ([(24, 8)], ["0::8"], None, {"distance tolerance": [2], "quality masking threshold": 20}, 1, 10),
    "masking_13_tolerance_0": ([(16, 6)], ["0:1:7"], None, {"distance tolerance": [0], "quality masking threshold": 13}, 1, 9),
    "two_segments": ([(12, 8), (10, 6)], ["0::8", "1:2:8"], None, {"distance tolerance": [1, 2]}, 2, 10),
    "two_segments_masked": ([(12, 8), (10, 6)], ["0::8", "1::6"], None, {"distance tolerance": [2, 1], "quality masking threshold": 10}, 2, 8),
    "reverse_complement": ([(24, 8)], ["0:-8:"], ["~0"], {"distance tolerance": [1]}, 1, 12),
    "knitted_two_tokens_in_one_segment": ([(20, 10)], ["0::4", "1:3:9"], ["0:~1"], {"distance tolerance": [2]}, 2, 9),
    "three_segments": ([(6, 6), (5, 7), (4, 5)], ["0::6", "1::7", "2:1:6"], None, {"distance tolerance": [1, 1, 1]}, 3, 8),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_the_oracle_and_the_second_restatement_agree(name):
    shapes, tokens, knit, extra, n_input, read_length = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31) if False else sum(map(ord, name)))
    segments = [synthetic.barcode_set(rng, count, length, min_distance=2 if count > 20 and length < 8 else 3) for count, length in shapes]
    spec = {"input segment cardinality": n_input, "sample": synthetic.decoder_spec("mdd", segments, tokens, knit)}
    job = capi.compile_job(spec)
    decoder = job["sample"]
    decoder.update(extra)              # into the COMPILED document: the compiler refuses a tolerance past the Shannon bound (metric.h:216-242), the decoder takes what it is given
    lengths = [int(x) for x in decoder["barcode length"]]
    tolerance = [int(x) for x in decoder["distance tolerance"]]
    masking = int(decoder.get("quality masking threshold", 0))
    assert tolerance == extra["distance tolerance"] and masking == extra.get("quality masking threshold", 0)
    rule, reads = draw_reads(rng, spec["sample"], decoder, lengths, READS, n_input, read_length)
    records = sorted(decoder["codec"].values(), key=lambda r: r["index"])
    barcodes = np.array([[second.ASCII_TO_BAM[c] for c in "".join(r["barcode"])] for r in records], dtype=np.uint8)
    index_of = np.array([0] + [int(r["index"]) for r in records], dtype=np.int32)

    decoded, distance = second.classify(barcodes, lengths, tolerance, masking, [second.observe(rule, read) for read in reads])

    engine = pyoracle.OracleJob(job)
    base, qual = engine.stage(reads)
    result, word = engine.classify(base, qual)
    got = result[:, 0]
    assert np.array_equal(got["index"], index_of[decoded]), name
    assert np.array_equal(got["distance"].astype(np.int64), distance), name
    assert np.array_equal(got["qcfail"] != 0, decoded == 0), name                     # mdd.cpp:82-84
    # every case decides something both ways and corrects errors: the agreement is not one of empty sets
    classified = decoded > 0
    assert 0.3 < classified.mean() < 0.999, (name, classified.mean())
    if max(tolerance) > 0:
        assert (distance[classified] > 0).mean() > 0.02, name
    else:
        assert not distance.any()                   # only the exact match can classify; a masked position does not stop that one (mdd.cpp:43-46)
