"""Quality control accumulators of the output reads (reference multiplex.h:28-181, multiplex.cpp:26-262,405-431): a job with
`enable quality control` runs FASTQ -> SAM; the report's "multiplex" array is held against a numpy restatement of the
reference's accumulators computed from the SAM records themselves (channel by read group, every output segment's bases and
qualities by cycle).  The reference holds no golden file for this block: parity unpinned, semantics from its source."""
import json
import os

import numpy as np
import pytest

import bench
from pheniqs_amd import DecoderSet, capi, synthetic

pytestmark = pytest.mark.gpu

BAM = {c: k for k, c in enumerate("=ACMGRSVTWYHKDBN")}
RANGE = 42


def quantile(distribution, count, portion):
    """NucleotideAccumulator::quantile, multiplex.h:99-113"""
    position = int(portion * count)
    phred = 0
    while position > 0:
        if distribution[phred] >= position:
            break
        position -= distribution[phred]
        phred += 1
        while phred < RANGE - 1 and distribution[phred] == 0:
            phred += 1
        if phred >= RANGE - 1:
            break
    return phred


def finalize(distribution):
    """NucleotideAccumulator::finalize, multiplex.cpp:77-104"""
    count = int(distribution.sum())
    out = dict(count=count, Q1=0, Q3=0, IQR=0, LW=0, RW=0, min=0, max=0, mean=0.0, median=0)
    if count:
        total = 0
        for q in range(RANGE):
            value = int(distribution[q])
            total += value * q
            if value:
                out["max"] = q
                if out["min"] == 0:
                    out["min"] = q
        out["mean"] = total / count
        out["median"] = quantile(distribution, count, 0.5)
        out["Q1"] = quantile(distribution, count, 0.25)
        out["Q3"] = quantile(distribution, count, 0.75)
        out["IQR"] = out["Q3"] - out["Q1"]
        w = out["Q1"] - out["IQR"] * 1.5
        out["LW"] = out["min"] if w < out["min"] else int(w)
        w = out["Q3"] + out["IQR"] * 1.5
        out["RW"] = out["max"] if w > out["max"] else int(w)
    return out


NAMES = {"count": "cycle count", "Q1": "cycle quality first quartile", "Q3": "cycle quality third quartile", "IQR": "cycle quality interquartile range",
         "LW": "cycle quality left whisker", "RW": "cycle quality right whisker", "min": "cycle quality min", "max": "cycle quality max",
         "mean": "cycle quality mean", "median": "cycle quality median"}


def test_multiplex_block_equals_the_accumulators_of_the_written_reads(tmp_path):
    n = 30000
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}           # two output segments: the template read and the index read
    spec["enable quality control"] = True
    job = capi.compile_job(spec)
    assert job["sample"]["enable quality control"] is True
    code, quality, flag, _ = synthetic.generate_rows(job, n, 3)
    letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    rng = np.random.default_rng(9)
    template = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, size=(n, 30), p=[0.245, 0.245, 0.245, 0.245, 0.02])]
    template_quality = (rng.choice([2, 11, 25, 37, 40], size=(n, 30)) + 33).astype(np.uint8)
    paths = []
    for k, (text, qual) in enumerate(((template, template_quality), (letters[code], (np.minimum(quality, 93) + 33).astype(np.uint8)))):
        path = str(tmp_path / ("segment_%d.fastq" % k))
        bench.write_fastq(path, k + 1, text, qual, flag)
        paths.append(path)
    s = DecoderSet(spec, device=0)
    s.stats_reset()
    out = str(tmp_path / "out.sam")
    incoming = s.transcode_fastq_to_sam(paths, out, 1 << 13)
    assert incoming[0] == n
    report = json.loads(s.report(s.stats_export(), incoming))
    multiplex = report["multiplex"]
    channels = 1 + len(job["sample"]["codec"])
    assert len(multiplex) == channels

    # the accumulators from the SAM records: channel by read group, segment by the first / last in template flags
    id_to_channel = {job["sample"]["undetermined"]["ID"]: 0}
    for ordinal, record in enumerate(job["sample"]["codec"].values()):
        id_to_channel[record["ID"]] = ordinal + 1
    cycles = [30, 8]
    by_cycle = [np.zeros((channels, cycles[k], 16, RANGE), dtype=np.int64) for k in range(2)]
    nucleic = np.zeros((channels, 2, 16), dtype=np.int64)
    mean_sum = np.zeros((channels, 2))
    mean_count = np.zeros((channels, 2), dtype=np.int64)
    mean_max = np.zeros((channels, 2))
    mean_distribution = np.zeros((channels, 2, RANGE), dtype=np.int64)
    records = 0
    for line in open(out):
        if line.startswith("@"):
            continue
        field = line.rstrip("\n").split("\t")
        segment = 0 if int(field[1]) & 64 else 1
        tags = dict(t.split(":", 2)[::2] for t in field[11:])
        channel = id_to_channel[tags["RG"]]
        codes = np.array([BAM[c] for c in field[9]])
        values = np.frombuffer(field[10].encode(), dtype=np.uint8).astype(np.int64) - 33
        np.add.at(by_cycle[segment][channel], (np.arange(len(codes)), codes, np.minimum(values, RANGE - 1)), 1)
        np.add.at(nucleic[channel, segment], codes, 1)
        nucleic[channel, segment, 0] += len(codes)
        value = values.sum() / len(values)
        mean_sum[channel, segment] += value
        mean_count[channel, segment] += 1
        mean_max[channel, segment] = max(mean_max[channel, segment], value)
        mean_distribution[channel, segment, min(int(value), RANGE - 1)] += 1
        records += 1
    assert records == 2 * n                                      # no outgoing filter: every read is written, both segments

    checked = 0
    for channel in range(channels):
        for segment in range(2):
            got = multiplex[channel]["quality control by segment"][segment]
            if mean_count[channel, segment] == 0:
                assert got["max sequence length"] == 0 and got["min sequence length"] == 0
                continue
            assert got["max sequence length"] == cycles[segment] and got["min sequence length"] == cycles[segment]
            by_code = by_cycle[segment][channel]                 # [cycle][code][phred]
            everything = by_code.sum(axis=1)                     # CycleAccumulator::finalize: code 0 holds every nucleotide
            expected = [finalize(everything[c]) for c in range(cycles[segment])]
            block = got["quality control by cycle"]["cycle quality distribution"]
            for key, name in NAMES.items():
                mine = [e[key] for e in expected]
                if key == "mean":
                    assert np.allclose(block[name], mine, rtol=0, atol=1e-12), (channel, segment, name)
                else:
                    assert block[name] == mine, (channel, segment, name)
            present = [c for c in range(1, 16) if nucleic[channel, segment, c] > 0]
            assert [e["nucleotide"] for e in got["quality control by nucleotide"]] == ["=ACMGRSVTWYHKDBN"[c] for c in present]
            for entry, c in zip(got["quality control by nucleotide"], present):
                assert entry["nucleotide count"] == int(nucleic[channel, segment, c])
                expected = [finalize(by_code[cycle, c]) for cycle in range(cycles[segment])]
                for key, name in NAMES.items():
                    mine = [e[key] for e in expected]
                    if key == "mean":
                        assert np.allclose(entry["cycle quality distribution"][name], mine, rtol=0, atol=1e-12)
                    else:
                        assert entry["cycle quality distribution"][name] == mine, (channel, segment, c, name)
            average = got["average phred score report"]
            assert average["average phred score distribution"] == [int(v) for v in mean_distribution[channel, segment]]
            assert average["average phred score min"] == 0.0          # the reference's minimum starts at 0 and stays there
            assert abs(average["average phred score max"] - mean_max[channel, segment]) < 1e-12
            assert abs(average["average phred score mean"] - mean_sum[channel, segment] / mean_count[channel, segment]) < 1e-6
            checked += 1
    assert checked > 150

    # the flat image: merging it into a second set doubles every count (Multiplexer::collect, multiplex.cpp:405-411)
    image = s.qc_export()
    other = DecoderSet(spec, device=capi.DEVICE_NONE)
    other.qc_merge(image)
    other.qc_merge(image)
    twice = other.qc_export()
    assert twice.size == image.size and int(twice[0]) == channels and int(twice[1]) == 2
    s.stats_reset()
    assert s.qc_export().size == 0 or int(s.qc_export()[0]) == 0
    s.close()
    other.close()
