"""The reference's own prior estimation test (reference test/api/prior): BDGGG_annotated.json + the report of a run
(BDGGG_annotated_report.json) must give valid/BDGGG_annotated_estimated.json.

tests/golden/prior holds those four files as data.  The reference's tool copies `estimated concentration` /
`estimated noise` from the report into the configuration (tool/pheniqs-prior-api.py:39-56); here the estimates are
recomputed (phq_report_json / phq_adjust_prior = Classifier::finalize, classifier.h:94-160) from the run's integer
counters, which the report states: count, pf count, low (conditional) confidence count per barcode.  Both the report's
estimates and the adjusted configuration must come out as the reference's files have them, at their 15 decimals.
"""
import json
import os

import numpy as np

from pheniqs_amd import DecoderSet, capi

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prior")


def _words_from_report(s, report):
    """The counter buffer (include/pheniqs_amd.h) of the run the report describes, as far as the estimator reads it."""
    words = np.zeros(s.stats_words, dtype=np.uint64)
    outgoing = report.get("outgoing", report["sample"])       # this report has no outgoing block: every read reaches the sample decoder
    words[0] = outgoing["count"] - outgoing["pf count"]
    words[1] = outgoing["pf count"]
    nodes = [report["sample"]] + list(report.get("molecular", [])) + list(report.get("cellular", []))
    assert len(nodes) == len(s.decoders)
    for d, node in zip(s.decoders, nodes):
        options = [node["unclassified"]] + sorted(node.get("classified", []), key=lambda o: o["index"])
        for b, o in enumerate(options):
            assert o["index"] == b
            fail = d["stats_offset"] + b * 12
            passed = fail + 6
            words[passed] = o["pf count"]
            words[fail] = o["count"] - o["pf count"]
            words[fail + 4] = o.get("low conditional confidence count", 0)
            words[fail + 5] = o.get("low confidence count", 0)
            # distances and confidences as the averages state them: not read by the estimator
            words[passed + 1] = int(round(o.get("average pf distance", 0) * o["pf count"]))
            words[fail + 1] = int(round(o.get("average distance", 0) * o["count"])) - int(words[passed + 1])
    return words


def test_import_is_resolved_by_the_library():
    """BDGGG_annotated.json imports BDGGG_interleave.json (job.cpp:163-224): what the imported document defines fills
    in what the importing one does not, the importing document wins."""
    job = capi.load_job(os.path.join(HERE, "BDGGG_annotated.json"))
    annotated = json.load(open(os.path.join(HERE, "BDGGG_annotated.json")))
    interleave = json.load(open(os.path.join(HERE, "BDGGG_interleave.json")))
    assert "import" not in job
    assert set(job) == (set(annotated) | set(interleave)) - {"import"}
    for key, value in annotated.items():
        if key != "import" and not isinstance(value, dict):
            assert job[key] == value
    for key, value in interleave.items():
        if key not in annotated:
            assert job[key] == value
    assert job["sample"]["codec"]["@AGGCAGAA"] == annotated["sample"]["codec"]["@AGGCAGAA"]
    assert job["decoder"] == interleave["decoder"]
    # the same document handed over as text, from the directory the import is relative to
    here = os.getcwd()
    try:
        os.chdir(HERE)
        compiled = capi.compile_job(annotated)
    finally:
        os.chdir(here)
    assert compiled["sample"]["codec"]["@AGGCAGAA"]["barcode"] == ["AGGCAGAA"]


def test_prior_estimation_reproduces_the_reference_files():
    job = capi.load_job(os.path.join(HERE, "BDGGG_annotated.json"))
    report = json.load(open(os.path.join(HERE, "BDGGG_annotated_report.json")))
    golden = json.load(open(os.path.join(HERE, "BDGGG_annotated_estimated.json")))
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    words = _words_from_report(s, report)

    # the report written from those counters states the reference's estimates
    mine = json.loads(s.report(words, (report["incoming"]["count"], report["incoming"]["pf count"])))
    assert mine["sample"]["estimated noise"] == report["sample"]["estimated noise"]
    theirs = {tuple(o["barcode"]): o for o in report["sample"]["classified"]}
    for option in mine["sample"]["classified"]:
        reference = theirs[tuple(option["barcode"])]
        for key in ("count", "pf count", "estimated concentration", "pf pooled classified fraction", "pooled fraction", "pf fraction"):
            assert option[key] == reference[key], (option["barcode"], key)

    # ... and the configuration adjusted from them is the reference's estimated configuration
    adjusted = s.adjust_prior(words, job)
    assert adjusted["sample"]["noise"] == golden["sample"]["noise"]
    assert set(adjusted["sample"]["codec"]) == set(golden["sample"]["codec"])
    for key, record in golden["sample"]["codec"].items():
        assert adjusted["sample"]["codec"][key]["concentration"] == record["concentration"], key
        assert adjusted["sample"]["codec"][key]["barcode"] == record["barcode"]
        assert adjusted["sample"]["codec"][key]["LB"] == record["LB"]
    for key in ("confidence threshold", "algorithm", "transform"):
        assert adjusted["sample"][key] == golden["sample"][key], key
    for key in ("CN", "DT", "PI", "PL", "PM", "SM", "flowcell id", "template", "input"):
        assert adjusted[key] == golden[key], key
    assert "decoder" not in adjusted and "base" not in adjusted["sample"]
    s.close()
