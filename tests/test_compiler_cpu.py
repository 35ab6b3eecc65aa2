"""Next row f4: the library's configuration compiler (csrc/compile.cpp, phq_compile_job) for the decoder topics.

Pinned to the reference's own compiled ontology of test/BDGGG (valid/compile_annotated.out, committed as
tests/golden/BDGGG/compile_annotated.json) and cross checked against the independent Python restatement the
tests and bench use (tests/jobspec.py).  No GPU.
"""
import copy
import json

import numpy as np
import pytest

import bdggg
from pheniqs_amd import DecoderSet, PheniqsError, capi, synthetic
import jobspec


def test_reference_compile_of_bdggg_is_reproduced():
    """valid/compile_annotated.out: every decoder member the classification path reads."""
    reference = bdggg.compiled_reference_job()
    ours = capi.compile_job(bdggg.raw_job())          # decoder repository and `base` references included
    assert "decoder" not in ours and ours["input segment cardinality"] == reference["input segment cardinality"] == 3
    for kind in ("sample", "molecular", "cellular"):
        theirs = reference[kind] if isinstance(reference[kind], list) else [reference[kind]]
        mine = ours[kind] if isinstance(ours[kind], list) else [ours[kind]]
        assert len(theirs) == len(mine)
        for t, m in zip(theirs, mine):
            for key in ("algorithm", "barcode length", "segment cardinality", "nucleotide cardinality", "index",
                        "high quality threshold", "high quality distance threshold", "quality masking threshold",
                        "confidence threshold", "noise", "corrected quality", "transform"):
                assert t[key] == m[key], (kind, key)
            # the reference prints 15 truncated decimals: 1/4^8 = 0.0000152587890625 -> 0.000015258789062
            assert 0 <= m["random barcode probability"] - t["random barcode probability"] < 1e-15
            if "codec" in t:
                assert t["distance tolerance"] == m["distance tolerance"]
                assert t["shannon bound"] == m["shannon bound"]
                assert t["barcode cardinality"] == m["barcode cardinality"]
                assert list(t["codec"].keys()) == list(m["codec"].keys())
                for key in t["codec"]:
                    # everything the reference's record holds but what the output compiler adds (TC, output)
                    for member in set(t["codec"][key]) - {"concentration", "TC", "output"}:
                        assert t["codec"][key][member] == m["codec"][key][member], (key, member)
                    assert set(m["codec"][key]) <= set(t["codec"][key])
                    delta = m["codec"][key]["concentration"] - t["codec"][key]["concentration"]
                    assert 0 <= delta < 1e-15 + 1e-18
            for member in set(t["undetermined"]) - {"TC", "output"}:
                assert t["undetermined"][member] == m["undetermined"][member], (kind, member)
            for member in set(t) - {"codec", "undetermined", "random barcode probability", "base output url", "output"}:
                assert t[member] == m[member], (kind, member)
    assert ours["sample"]["multiplexing classifier"] is True


def test_io_directives_of_the_reference_compile_are_reproduced():
    """valid/compile_annotated.out: `input` under `base input url`, the template's default knit and the output segment
    cardinality, and on every channel of the multiplexing decoder `TC` and one output URL per output segment
    (transcode.cpp:500-557,1261-1417,1419-1460).  The reference appends the canonical query (?format=..): paths compared."""
    reference = bdggg.compiled_reference_job()
    ours = capi.compile_job(bdggg.raw_job())
    path = lambda url: url.split("?")[0]
    assert [path(u) for u in ours["input"]] == [path(u) for u in reference["input"]]
    assert ours["template"] == reference["template"]
    assert ours["output segment cardinality"] == reference["output segment cardinality"] == 2
    records = [(reference["sample"]["undetermined"], ours["sample"]["undetermined"])]
    records += [(reference["sample"]["codec"][key], ours["sample"]["codec"][key]) for key in reference["sample"]["codec"]]
    for theirs, mine in records:
        assert mine["TC"] == theirs["TC"] == 2
        assert [path(u) for u in mine["output"]] == [path(u) for u in theirs["output"]] == ["/dev/stdout", "/dev/stdout"]
    # relative outputs go under the base, a single URL is repeated per output segment, several must match the count
    raw = bdggg.raw_job()
    raw["output"] = ["all.sam"]
    raw["sample"]["codec"]["@AGGCAGAA"]["output"] = ["a_1.fastq.gz", "/tmp/a_2.fastq.gz"]
    ours = capi.compile_job(raw)
    assert ours["sample"]["codec"]["@AGGCAGAA"]["output"] == ["test/BDGGG/a_1.fastq.gz", "/tmp/a_2.fastq.gz"]
    assert ours["sample"]["codec"]["@CGTACTAG"]["output"] == ["test/BDGGG/all.sam"] * 2
    assert ours["sample"]["undetermined"]["output"] == ["test/BDGGG/all.sam"] * 2
    raw["sample"]["codec"]["@AGGCAGAA"]["output"] = ["a", "b", "c"]
    with pytest.raises(PheniqsError) as caught:
        capi.compile_job(raw)
    assert "incorrect number of output URLs in channel" in str(caught.value)
    # a job without a template routes every input segment to the output verbatim
    raw = bdggg.raw_job()
    del raw["template"]
    ours = capi.compile_job(raw)
    assert ours["template"] == {"transform": {"token": ["0::", "1::", "2::"], "knit": ["0", "1", "2"]}}
    assert ours["output segment cardinality"] == 3 and ours["sample"]["undetermined"]["TC"] == 3


def _without_io(job):
    """The Python restatement stops at the decoder topics: drop what the IO step adds (covered by the test above)."""
    for key in ("output", "output segment cardinality", "template"):
        job.pop(key, None)
    for topic in ("sample", "molecular", "cellular"):
        for decoder in ([job[topic]] if isinstance(job.get(topic), dict) else job.get(topic, [])):
            if decoder.get("multiplexing classifier"):
                decoder.pop("output", None)
                for record in [decoder.get("undetermined", {})] + list(decoder.get("codec", {}).values()):
                    record.pop("output", None)
                    record.pop("TC", None)
    return job


@pytest.mark.parametrize("name,algorithm", [("C2", "pamld"), ("C3", "pamld"), ("C4", "mdd"), ("C5", "pamld")])
def test_equals_the_python_restatement(name, algorithm):
    spec = synthetic.config_spec(name, algorithm)
    assert _without_io(capi.compile_job(spec)) == jobspec.compile_job(spec)          # doubles bit for bit: round trip digits


def test_compiled_documents_pass_through_and_create_compiles_by_itself():
    spec = bdggg.user_job()
    compiled = capi.compile_job(spec)
    assert capi.compile_job(compiled) == compiled
    a = DecoderSet(spec, device=capi.DEVICE_NONE)                        # a user level document runs unchanged
    b = DecoderSet(compiled, device=capi.DEVICE_NONE)
    assert (a.n_decoder, a.n_result, a.plane_width, a.stats_words) == (b.n_decoder, b.n_result, b.plane_width, b.stats_words)
    assert a.decoders == b.decoders


def test_decoder_repository_and_base_references():
    """Transcode::apply_inheritance (transcode.cpp:329-437): members of the base fill in what is not defined."""
    codec = {"@" + w: {"barcode": [w]} for w in ("AGGCAGAA", "CGTACTAG", "GGACTCCT")}
    spec = {
        "input": ["r1.fastq", "i1.fastq", "r2.fastq"],
        "decoder": {
            "root": {"noise": 0.02, "confidence threshold": 0.9, "transform": {"token": ["1::8"]}},
            "index8": {"base": "root", "codec": codec, "confidence threshold": 0.99},
            "unused": None,
        },
        "sample": {"base": "index8", "algorithm": "pamld", "high quality threshold": 20},
        "cellular": [{"base": "index8", "algorithm": "mdd", "noise": 0.05}],
    }
    job = capi.compile_job(spec)
    assert "decoder" not in job and job["input segment cardinality"] == 3
    sample, cellular = job["sample"], job["cellular"][0]
    assert "base" not in sample and "base" not in cellular
    assert (sample["noise"], sample["confidence threshold"], sample["high quality threshold"]) == (0.02, 0.99, 20)
    assert (cellular["noise"], cellular["algorithm"], cellular["index"]) == (0.05, "mdd", 0)
    assert sample["transform"] == {"token": ["1::8"], "knit": ["0"]}
    assert [r["index"] for r in sample["codec"].values()] == [1, 2, 3]
    assert sum(r["concentration"] for r in sample["codec"].values()) == pytest.approx(0.98, abs=1e-15)
    assert sum(r["concentration"] for r in cellular["codec"].values()) == pytest.approx(0.95, abs=1e-15)
    # the explicit equivalent
    flat = {"input segment cardinality": 3,
            "sample": {"algorithm": "pamld", "noise": 0.02, "confidence threshold": 0.99, "high quality threshold": 20,
                       "transform": {"token": ["1::8"]}, "codec": copy.deepcopy(codec)},
            "cellular": [{"algorithm": "mdd", "noise": 0.05, "confidence threshold": 0.99,
                          "transform": {"token": ["1::8"]}, "codec": copy.deepcopy(codec)}]}
    expected = jobspec.compile_job(flat)
    job = _without_io(job)
    assert job["sample"] == expected["sample"] and job["cellular"] == expected["cellular"]


@pytest.mark.parametrize("mutate,text", [
    (lambda s: s["sample"]["codec"].update({"@dup": {"barcode": ["AGGCAGAA"]}}), "duplicate barcode sequence AGGCAGAA"),
    (lambda s: s["sample"]["codec"].update({"@short": {"barcode": ["ACGT"]}}), "expected 8 but found 4 nucleotides in segment 0 of barcode @short"),
    (lambda s: s["sample"]["codec"].update({"@two": {"barcode": ["ACGTACGT", "ACGT"]}}), "expected 1 segments in barcode @two"),
    (lambda s: s["sample"].update({"distance tolerance": [3]}), "barcode tolerance for segment 0 is higher than shannon bound"),
    (lambda s: s["sample"].update({"distance tolerance": [1, 1]}), "distance tolerance cardinality inconsistant"),
    (lambda s: s["sample"].update({"random barcode probability": 1e-9}), "random barcode probability is smaller than lower bound"),
    (lambda s: s["sample"]["transform"].update({"token": ["1:8"]}), "illegal token syntax 1:8"),
    (lambda s: s["sample"]["transform"].update({"token": ["1:4:"]}), "is not fixed width"),
    (lambda s: s["sample"]["transform"].update({"token": ["7::8"]}), "invalid input feed reference 7 in token 0"),
    (lambda s: s["sample"]["transform"].update({"knit": ["0:3"]}), "invalid token reference 3 in transform"),
    (lambda s: s["sample"].update({"base": "nowhere"}) or s.update({"decoder": {}}), "reference to an unknown base nowhere"),
    (lambda s: s.update({"decoder": {"a": {"base": "a"}}}), "a references itself as parent"),
    (lambda s: [r.update({"concentration": 0}) for r in s["sample"]["codec"].values()], "total pool concentration is not a positive number"),
])
def test_configuration_errors_carry_the_reference_texts(mutate, text):
    spec = bdggg.user_job()
    mutate(spec)
    with pytest.raises(PheniqsError) as e:
        capi.compile_job(spec)
    assert e.value.code == 3 and text in str(e.value), str(e.value)
