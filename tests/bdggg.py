"""Loader for the reference's own decode test (reference test/BDGGG): inputs, config and golden outputs.

The files under tests/golden/BDGGG are DATA copied from the reference's test directory:
three FASTQ inputs, the two user level configs, and valid/{annotated.out, annotated.err,
compile_annotated.out} (renamed with .sam/.json suffixes).
"""
import copy
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "BDGGG")


def read_fastq(path, ascii_to_bam, phred_offset=33):
    """fastq.h:55-78 decode + fastq.h:253-285 Illumina comment filter flag."""
    records = []
    with open(path, "rb") as handle:
        lines = handle.read().split(b"\n")
    for i in range(0, len(lines) - 3, 4):
        head = lines[i][1:]
        name, _, comment = head.partition(b" ")
        sequence, quality = lines[i + 1], lines[i + 3]
        qcfail = False
        fields = comment.split(b":")
        if len(fields) > 1 and len(fields[1]) >= 1:
            qcfail = fields[1][:1] == b"Y"
        code = ascii_to_bam(sequence)
        qual = (np.frombuffer(quality, dtype=np.uint8) - phred_offset).astype(np.uint8)
        records.append({"name": name.decode(), "code": code, "quality": qual, "qcfail": qcfail,
                        "sequence": sequence.decode(), "quality_text": quality.decode()})
    return records


def raw_job():
    """BDGGG_annotated.json as the reference reads it: the file itself, its `import` of BDGGG_interleave.json resolved
    by the library (phq_load_job = Job::read_instruction_document, job.cpp:163-224); the decoder repository and the
    `base` references are left to the library's compiler."""
    from pheniqs_amd import capi
    return capi.load_job(os.path.join(HERE, "BDGGG_annotated.json"))


def user_job():
    """BDGGG_annotated.json with its import and `base` inheritance resolved by hand
    (Transcode::apply_inheritance transcode.cpp:328-443 is out of scope)."""
    annotated = json.load(open(os.path.join(HERE, "BDGGG_annotated.json")))
    interleave = json.load(open(os.path.join(HERE, "BDGGG_interleave.json")))
    repository = interleave["decoder"]

    def resolve(decoder):
        decoder = copy.deepcopy(decoder)
        base = decoder.pop("base", None)
        if base:
            merged = copy.deepcopy(repository[base])
            for key, value in decoder.items():
                if key == "codec":
                    for name, record in value.items():
                        merged.setdefault("codec", {}).setdefault(name, {}).update(record)
                else:
                    merged[key] = value
            decoder = merged
        return decoder

    job = {
        "input segment cardinality": 3,
        "min input length": annotated["min input length"],
        "filter incoming qc fail": interleave["filter incoming qc fail"],
        "leading segment index": 0,
        "sample": resolve(annotated["sample"]),
        "molecular": [resolve(d) for d in annotated["molecular"]],
        "cellular": [resolve(d) for d in annotated["cellular"]],
    }
    return job


def compiled_reference_job():
    return json.load(open(os.path.join(HERE, "compile_annotated.json")))


def golden_report():
    return json.load(open(os.path.join(HERE, "annotated.report.json")))


def golden_records():
    """First SAM record of every output read: name -> {flag, tags}."""
    out = {}
    order = []
    with open(os.path.join(HERE, "annotated.out.sam")) as handle:
        for line in handle:
            if line.startswith("@"):
                continue
            fields = line.rstrip("\n").split("\t")
            name, flag = fields[0], int(fields[1])
            if not flag & 64:
                continue
            tags = {}
            for field in fields[11:]:
                key, kind, value = field.split(":", 2)
                tags[key] = value
            out[name] = {"flag": flag, "tags": tags, "sequence": fields[9], "quality": fields[10]}
            order.append(name)
    return out, order


def load_reads(ascii_to_bam):
    segments = [read_fastq(os.path.join(HERE, "BDGGG_s0%d.fastq" % k), ascii_to_bam) for k in (1, 2, 3)]
    reads = []
    for records in zip(*segments):
        assert len({r["name"] for r in records}) == 1       # read.h:253-264 validate
        reads.append({"name": records[0]["name"], "segments": records, "qcfail": records[0]["qcfail"]})
    return reads
