"""tests/sam_restatement.py — the independent Python formatter of the reference's SAM records — is pinned here: with the
oracle's decisions for the reference's own test reads and the reference's own compiled job (tests/golden/BDGGG, data of the
reference's test directory) it must write the reference's golden file, test/BDGGG/valid/annotated.out, byte for byte.  The GPU
half (tests/test_gpu_record_device.py) then holds the library's device and host record writers against it on job shapes the
golden file has no case of.
"""
import os

import numpy as np
import pytest

import bdggg
import jobspec
import sam_restatement


def test_the_restatement_writes_the_golden_sam_file(oracle):
    job = bdggg.compiled_reference_job()
    files = [sam_restatement.read_fastq(os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k)) for k in (1, 2, 3)]
    assert len(files[0]) == 250
    # the oracle's decisions for the reads that pass the incoming filters (the oracle's own staging of the same text)
    engine = oracle.OracleJob(jobspec.compile_job(bdggg.user_job()))
    minimum = job["min input length"]
    kept = [k for k in range(250) if not any(len(files[i][k][2]) < minimum[i] for i in range(1, 3))]
    base, quality = engine.stage([[(files[i][k][2], files[i][k][3]) for i in range(3)] for k in kept])
    flags = np.array([1 if files[0][k][1] else 0 for k in kept], dtype=np.uint8)
    result, word, _, confidence = engine.classify(base, quality, flags, summary=True)
    dropped = []
    mine = sam_restatement.records(job, files, result, word, confidence, filtered=dropped)
    golden = b"".join(line for line in open(os.path.join(bdggg.HERE, "annotated.out.sam"), "rb").readlines() if not line.startswith(b"@"))
    assert len(dropped) == 2 and len(kept) == 248
    if mine != golden:
        for k, (a, b) in enumerate(zip(mine.split(b"\n"), golden.split(b"\n"))):
            assert a == b, (k, a, b)
    assert mine == golden


def test_tokens_follow_the_reference():
    """transform.h:65-80 on the cases the golden job has none of: offsets from the end, an end past the sequence, a start past it
    (which the reference reads as 0), an empty slice, `~`"""
    code = sam_restatement.ASCII_TO_BAM[np.frombuffer(b"ACGTTGCA", dtype=np.uint8)]
    quality = np.arange(8, dtype=np.uint8)
    cases = [("0:-3:", "GCA"), ("0:2:100", "GTTGCA"), ("0:9:", "ACGTTGCA"), ("0:5:3", ""), ("0:-100:2", "AC"), ("0:1:-1", "CGTTGC")]
    for pattern, expected in cases:
        token = sam_restatement.transforms_of({"token": [pattern]})[0]
        got, q = sam_restatement._slice(token, code, quality)
        assert "".join(sam_restatement.IUPAC[c] for c in got) == expected, pattern
        assert len(q) == len(got)
    token = sam_restatement.transforms_of({"token": ["0:1:4"], "knit": ["~0"]})[0]
    got, q = sam_restatement._slice(token, code, quality)
    assert "".join(sam_restatement.IUPAC[c] for c in got) == "ACG" and list(q) == [3, 2, 1]          # CGT reversed and complemented
