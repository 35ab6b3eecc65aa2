"""Parity of the HIP path against the CPU oracle and the reference's golden files, through the C ABI.

Bar (BASELINE.json north_star): decoded index, distance, high quality distance, status, QC fail, channel
word and every integer counter are bit exact; the PAMLD posterior is within 1e-6 of the reference
(measured ~1e-16, asserted at 1e-12 so that a regression in the in-kernel 2^x shows up long before the
contract is at risk); confidence sums agree to 1e-12 relative.
"""
import json
import os

import numpy as np
import pytest

import bdggg
from pheniqs_amd import DecoderSet, PheniqsError, RESULT_DTYPE, capi, synthetic
import jobspec

pytestmark = pytest.mark.gpu

POSTERIOR_TOLERANCE = 1e-6        # the contract
# what the kernels are designed to hold: the literal kernel differs from the oracle only through its 2^x
# (~1e-16); the fast mode's first tier finishes a read only when what it did not score is bounded by 1e-8 of the
# winner (the posterior is taken at the middle: 5e-9) and everything but the winner weighs < 5e-4 of it, summed in
# single precision (relative error ~1e-6: 5e-10); the later tiers hold 1e-9 and better
POSTERIOR_DESIGN_TOLERANCE = {capi.MODE_LITERAL: 1e-12, capi.MODE_FAST: 1e-8}
INTEGER_FIELDS = ("index", "distance", "hq_distance", "status", "qcfail")
INTEGER_COUNTERS = ("count", "pf_count", "accumulated_distance", "accumulated_pf_distance",
                    "low_conditional_confidence_count", "low_confidence_count")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def run_gpu(torch, job, code, quality, flag, offsets, mode=capi.MODE_LITERAL, pieces=1, device_set=None):
    """Stage rows into decoder planes on the GPU and classify through phq_classify_device."""
    s = device_set or DecoderSet(job, device=0)
    s.set_mode(mode)
    s.stats_reset()
    n = code.shape[0]
    base = torch.from_numpy(s.planes_from_rows(code, offsets)).cuda()
    qual = torch.from_numpy(s.planes_from_rows(quality, offsets)).cuda()
    flags = torch.from_numpy(np.ascontiguousarray(flag, dtype=np.uint8)).cuda() if flag is not None else None
    result = torch.zeros(max(1, s.n_result * n) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    word = torch.full((max(1, n),), -1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    bounds = np.linspace(0, n, pieces + 1).astype(np.int64)
    for first, last in zip(bounds[:-1], bounds[1:]):
        s.classify_device(n, int(first), int(last - first), base, qual, flags, result, word, stream)
    torch.cuda.synchronize()
    got = result.cpu().numpy().view(RESULT_DTYPE)[: s.n_result * n].reshape(s.n_result, n)
    return s, got, word.cpu().numpy().view(np.uint32)[:n], s.stats_export()


def assert_parity(oracle, torch, job, code, quality, flag, offsets, **kwargs):
    """Without an explicit mode the case runs in both: PHQ_MODE_LITERAL and PHQ_MODE_FAST (the default)."""
    if "mode" not in kwargs:
        out = None
        for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
            out = assert_parity(oracle, torch, job, code, quality, flag, offsets, mode=mode, **kwargs)
        return out
    engine = oracle.OracleJob(job)
    expected, expected_word = engine.classify(code, quality, flag)
    s, got, word, words = run_gpu(torch, job, code, quality, flag, offsets, **kwargs)
    n = code.shape[0]
    for slot in range(s.n_result):
        for name in INTEGER_FIELDS:
            mismatch = np.nonzero(got[slot][name] != expected[:, slot][name])[0]
            assert mismatch.size == 0, (slot, name, mismatch[:5], got[slot][mismatch[:5]], expected[mismatch[:5], slot])
        if n:
            delta = np.abs(got[slot]["confidence"] - expected[:, slot]["confidence"]).max()
            assert delta < POSTERIOR_TOLERANCE
            assert delta < POSTERIOR_DESIGN_TOLERANCE[kwargs.get("mode", capi.MODE_LITERAL)], delta
    assert np.array_equal(word, expected_word)
    count, pf_count = engine.outgoing()
    assert (int(words[0] + words[1]), int(words[1])) == (count, pf_count)
    for k in range(s.n_decoder):
        mine, theirs = s.stats_options(words, k), engine.options(k)
        for name in INTEGER_COUNTERS:
            assert np.array_equal(mine[name], theirs[name]), (k, name)
        tolerance = 1e-12 if kwargs.get("mode", capi.MODE_LITERAL) == capi.MODE_LITERAL else 1e-8
        for name in ("accumulated_confidence", "accumulated_pf_confidence"):
            assert np.allclose(mine[name], theirs[name], rtol=tolerance, atol=tolerance), (k, name)
    return s, engine, got, words


# ---- the reference's own decode test ----------------------------------------------------------------------


def test_bdggg_golden_through_the_gpu(oracle, torch_cuda):
    """test/BDGGG end to end: FASTQ -> phq_stage_reads -> kernels -> report, against valid/annotated.{out,err}."""
    job = jobspec.compile_job(bdggg.user_job())
    reads = bdggg.load_reads(oracle.ascii_to_bam)
    s = DecoderSet(job, device=0)
    segments = []
    for k in range(3):
        codes = [r["segments"][k]["code"] for r in reads]
        length = np.array([len(c) for c in codes], dtype=np.int32)
        segments.append({"code": np.concatenate(codes), "quality": np.concatenate([r["segments"][k]["quality"] for r in reads]),
                         "offset": np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int64), "length": length})
    flags = np.array([1 if r["qcfail"] else 0 for r in reads], dtype=np.uint8)
    staged = s.stage_reads(segments, flags)
    n = staged["n"]
    assert n == 248 and staged["capacity"] == 250
    result, word = s.classify_host(n, staged["base"], staged["quality"], staged["flag"], capacity=staged["capacity"])
    words = s.stats_export()

    records, order = bdggg.golden_records()
    kept = [reads[i] for i in staged["kept"]]
    assert [r["name"] for r in kept] == order
    id_to_index = {"undetermined": 0}
    for record in job["sample"]["codec"].values():
        id_to_index[record["barcode"][0]] = record["index"]
    for i, read in enumerate(kept):
        tags = records[read["name"]]["tags"]
        assert id_to_index[tags["RG"].split(":")[-1]] == result[0][i]["index"], read["name"]
        assert bool(records[read["name"]]["flag"] & 512) == bool(word[i] >> 31), read["name"]
        for tag, slot in (("XB", 0), ("XC", 1)):
            confidence = result[slot][i]["confidence"]          # one decoder per type: the type product is the value itself
            if 0 < confidence < 1:
                assert tags[tag] == "%g" % float(np.float32(1.0 - confidence)), (read["name"], tag)
            else:
                assert tag not in tags

    report = json.loads(s.report(words, staged["incoming"]))
    golden = bdggg.golden_report()

    def compare(mine, theirs, path):
        if isinstance(mine, dict):
            for key, value in mine.items():
                assert key in theirs, path + "/" + key
                compare(value, theirs[key], path + "/" + key)
        elif isinstance(mine, list):
            assert len(mine) == len(theirs), path
            for k, (m, t) in enumerate(zip(mine, theirs)):
                compare(m, t, "%s[%d]" % (path, k))
        elif isinstance(theirs, float):
            # 15 truncated decimals in the golden file; our posterior differs from the reference's by ~1e-14
            assert abs(mine - theirs) <= 1e-12, (path, mine, theirs)
        else:
            assert mine == theirs, (path, mine, theirs)

    compare(report, golden, "")
    assert set(report.keys()) == set(golden.keys())


@pytest.mark.parametrize("name", ["H94", "U384", "S384", "L16"])
def test_tied_noise_reads_are_charged_to_the_references_argmax(oracle, torch_cuda, name):
    """Half of the reads noise, a third to a half of them tied at the top in single precision: the reference charges a noise read
    to its argmax (`low conditional confidence count`, pamld.cpp:113-119), and which of two barcodes at the same distance with the
    same qualities under their mismatches that is, is a matter of the last bits of its own Kahan sums.  The product form kernel
    names the two blocks of eight barcodes that hold every candidate and the exact tier's first pass walks them literally
    (pamld_lut.hpp NEAR_TOP, pamld_exact_lanes.hip TIE_PACE); ties over more blocks take the scan.  Every integer — the per barcode
    counts among them — must be the oracle's; with `PHQ_NO_TIES=1` the same reads take the scan, to the same counters."""
    spec = synthetic.config_spec(name, "pamld")
    spec["sample"]["noise"] = 0.5
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 16, 77)
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)
    options = engine.options(0)
    assert options["low_conditional_confidence_count"].sum() > 20000 and (options["low_conditional_confidence_count"][1:] > 0).sum() > 50


# ---- BASELINE configs at oracle-sized samples ---------------------------------------------------------------


MODES = [capi.MODE_LITERAL, capi.MODE_FAST]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,algorithm,n", [
    ("C2", "pamld", 1 << 16), ("C2", "mdd", 1 << 16),
    ("C3", "pamld", 3000), ("C3", "mdd", 1 << 14),
    ("C4", "pamld", 1 << 14), ("C4", "mdd", 1 << 14), ("C4V", "pamld", 1 << 14), ("C4V", "mdd", 1 << 14),
    ("C5", "pamld", 1 << 14), ("C5", "mdd", 1 << 14),
    # the reference's own published workload (docs/illumina_api_vignette.md:222): 94 wells of a 12 x 8 dual index plate
    ("H94", "pamld", 1 << 16), ("H94A", "pamld", 1 << 16), ("H94", "mdd", 1 << 14),
])
def test_baseline_configs_against_oracle(oracle, torch_cuda, name, algorithm, n, mode):
    job, recipe = synthetic.config_job(name, algorithm)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)


@pytest.mark.parametrize("rows,columns,lengths,wells_missing,spread,noise", [
    (12, 8, (8, 8), 2, 0.0, 0.05), (12, 8, (8, 8), 0, 0.0, 0.01), (16, 16, (8, 8), 40, 0.0, 0.01), (12, 8, (8, 8), 2, 1.0, 0.02),
    (5, 9, (6, 7), 11, 0.5, 0.01), (3, 16, (8, 5), 0, 2.0, 0.01), (2, 2, (4, 4), 1, 0.3, 0.1), (16, 12, (7, 8), 100, 1.0, 0.01),
    # round 6: 171 to 256 wells — the bins do not fit the workgroup, the classified reads' counters come from the records
    (14, 14, (8, 8), 0, 0.0, 0.02), (16, 16, (6, 7), 0, 0.0, 0.01), (16, 15, (7, 6), 20, 1.0, 0.02), (16, 12, (6, 6), 0, 0.5, 0.01)])
def test_plates_every_word_of_both_segments_scored(oracle, torch_cuda, rows, columns, lengths, wells_missing, spread, noise):
    """Dual index plates (pamld_plate.hip): few words a segment, every combination or a part of them, uniform priors or
    priors that differ; the wells that are missing still get reads (sequencing errors, and draws put ON them here).  The
    kernel must take the set, finish most reads, and everything must be the oracle's in both modes."""
    rng = np.random.default_rng(1000 * rows + 10 * columns + wells_missing)
    first = synthetic.barcode_set(rng, rows, lengths[0], min_distance=3 if lengths[0] >= 6 else 2)
    second = synthetic.barcode_set(rng, columns, lengths[1], min_distance=3 if lengths[1] >= 6 else 2)
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("pamld", [first, second], ["0::%d" % lengths[0], "1::%d" % lengths[1]], noise=noise)}
    keys = list(spec["sample"]["codec"])
    missing = set(rng.choice(len(keys), size=wells_missing, replace=False).tolist()) if wells_missing else set()
    wells = [(key, spec["sample"]["codec"][key]) for k, key in enumerate(keys) if k not in missing]
    spec["sample"]["codec"] = {key: dict(record, concentration=float(10 ** rng.uniform(-spread, spread)) if spread else 1) for key, record in wells}
    job = capi.compile_job(spec)
    n = 1 << 16
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 17 + rows)
    if missing:          # reads ON wells that are not in the set
        letters = {"A": 1, "C": 2, "G": 4, "T": 8}
        picks = rng.choice(sorted(missing), size=1 << 9)
        code[: 1 << 9] = np.array([[letters[c] for c in first[k // columns] + second[k % columns]] for k in picks], dtype=np.uint8)
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    # (a set of some two hundred barcodes may lose its workgroup histogram to the product form kernel's occupancy, and with it
    # this tier: the chain is asserted where the histogram is sure to stay)
    if max(rows, columns, 8) < rows * columns - wells_missing <= 128:          # (up to 8 barcodes are the few barcode kernel's)
        s.set_mode(capi.MODE_FAST)
        s.profile_enable(True)
        run_gpu(torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST, device_set=s)
        chain = [span["kernel"] for span in s.profile_chain()]
        assert chain[0] == "pamld_plate_kernel", chain
    if rows * columns - wells_missing > 200 and min(lengths) < 8:
        assert [c["kernel"] for c in s.plan_chain(n, capi.MODE_FAST)][0] == "pamld_plate_kernel"
    assert int((got[0]["status"] == 0).sum()) > n // 2
    s.close()


@pytest.mark.parametrize("length,count,spread,absent", [(8, 96, 1.0, 0), (8, 96, 3.0, 0), (10, 96, 1.0, 0), (6, 40, 1.5, 0), (9, 64, 2.0, 0), (8, 192, 1.0, 0),
                                                        (5, 20, 0.5, 0), (8, 96, 1.0, 5), (10, 64, 2.0, 2),
                                                        # round 6: the 512 lane layout with prior ratios (193 to ~850 barcodes of up to 8 nt)
                                                        (8, 384, 1.0, 0), (8, 512, 2.0, 7), (7, 300, 1.0, 0), (8, 800, 0.5, 0)])
def test_neighbour_lists_with_priors(oracle, torch_cuda, length, count, spread, absent):
    """Single index sets whose priors differ (what phq_adjust_prior leaves behind): the neighbour list kernel weighs the
    listed barcodes and its bound by their priors over the nearest barcode's (pamld.cpp:66-79); 192 barcodes with their
    weights do not fit a CU's LDS and stay with the product form kernel.  `absent`: every absent-th barcode has prior
    zero; reads still fall on it (sequencing errors of its neighbours, the generator's own draws are replaced)."""
    rng = np.random.default_rng(700 + 10 * length + count)
    words = synthetic.barcode_set(rng, count, length, min_distance=3 if length >= 6 else 2)
    spec = {"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words], ["0::%d" % length])}
    for k, record in enumerate(spec["sample"]["codec"].values()):
        record["concentration"] = 0 if (absent and k % absent == 1) else float(10 ** rng.uniform(-spread, spread))
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 16, 77 + length)
    if absent:          # reads ON the barcodes whose prior is zero
        picks = rng.integers(0, count, size=1 << 10)
        code[: 1 << 10] = np.array([[{"A": 1, "C": 2, "G": 4, "T": 8}[c] for c in words[k]] for k in picks], dtype=np.uint8)
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)
    assert int((got[0]["status"] == 0).sum()) > (1 << 15)
    if count > 288:
        assert [c["kernel"] for c in s.plan_chain(1 << 16, capi.MODE_FAST)][0] == "pamld_near_kernel"


@pytest.mark.parametrize("priors", [False, True])
@pytest.mark.parametrize("length,count", [(1, 2), (2, 4), (3, 8), (4, 5), (5, 1), (6, 4), (7, 8), (8, 3), (8, 8)])
def test_codecs_of_a_few_barcodes(oracle, torch_cuda, length, count, priors):
    """At most 8 barcodes of up to 8 nt (a SPLiT-seq round, a multiplex of a handful of samples) take the kernel of
    pamld_few.hip in fast mode: every barcode scored in fp64 by masks, the posterior by a refined reciprocal, bins in
    16 copies.  As first decoder (with and without an input flag plane) and behind another one (the state word), with
    bases outside ACGT, qualities of 0 and of 128 and more, priors that differ and a prior of zero; the literal mode
    beside it.  Held to 1e-12 like pamld_tiny.hip, whose arithmetic it is."""
    rng = np.random.default_rng(9000 + 10 * length + count + (100 if priors else 0))
    words = synthetic.barcode_set(rng, count, length, min_distance=1 if length < 6 else 2)
    front = synthetic.barcode_set(rng, 5, 8, min_distance=3)
    spec = {"input segment cardinality": 2,
            "cellular": [synthetic.decoder_spec("pamld", [front], ["1::8"]), synthetic.decoder_spec("pamld", [words], ["0::%d" % length])],
            "sample": synthetic.decoder_spec("pamld", [words], ["0::%d" % length])}
    if priors:
        for decoder in (spec["sample"], spec["cellular"][1]):
            for k, record in enumerate(decoder["codec"].values()):
                record["concentration"] = 0 if (count > 2 and k == 1) else float(10 ** rng.uniform(-1.0, 1.0))
    job = capi.compile_job(spec)
    n = 1 << 14
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 31 + length)
    code[rng.random(code.shape) < 0.01] = 15
    quality[rng.random(quality.shape) < 0.01] = 0
    quality[rng.random(quality.shape) < 0.002] = 200
    flag[:] = rng.random(n) < 0.1
    for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
        s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)
        expected, _ = oracle.OracleJob(job).classify(code, quality, flag)
        for slot in range(s.n_result):         # all three are codecs of a few barcodes: first in the chain and behind another decoder
            assert float(np.abs(got[slot]["confidence"] - expected[:, slot]["confidence"]).max()) <= 1e-12
    assert_parity(oracle, torch_cuda, job, code, quality, None, offsets, mode=capi.MODE_FAST)


@pytest.mark.parametrize("lengths,count,spread,absent", [((8, 8), 96, 1.0, 0), ((10, 10), 96, 1.0, 0), ((10, 10), 196, 2.0, 0), ((6, 9), 48, 0.5, 0),
                                                         ((8, 8), 96, 1.0, 4), ((10, 10), 64, 0.5, 2)])
def test_paired_sets_with_priors(oracle, torch_cuda, lengths, count, spread, absent):
    """Unique dual index / ligation pairs whose priors differ: the paired kernel scales its bound by the other barcodes'
    priors over the named one's and takes the noise term over that prior (pamld.cpp:29,66-86)."""
    rng = np.random.default_rng(900 + lengths[0] + count)
    first, second = synthetic.barcode_set(rng, count, lengths[0]), synthetic.barcode_set(rng, count, lengths[1])
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("pamld", [first, second], ["0::%d" % lengths[0], "1::%d" % lengths[1]], pairing="zip")}
    for k, record in enumerate(spec["sample"]["codec"].values()):
        record["concentration"] = 0 if (absent and k % absent == 1) else float(10 ** rng.uniform(-spread, spread))
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 15, 91 + count)
    if absent:          # reads ON the pairs whose prior is zero
        picks = rng.integers(0, count, size=1 << 10)
        code[: 1 << 10] = np.array([[{"A": 1, "C": 2, "G": 4, "T": 8}[c] for c in first[k] + second[k]] for k in picks], dtype=np.uint8)
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)
    assert int((got[0]["status"] == 0).sum()) > (1 << 14)


@pytest.mark.parametrize("length,count,min_distance,tolerance,masking", [
    (8, 96, 3, None, 0), (8, 96, 3, [0], 0), (8, 96, 3, None, 20), (6, 30, 2, None, 12), (5, 12, 3, None, 0), (7, 64, 3, None, 30),
    (9, 96, 3, None, 0), (10, 200, 3, None, 25), (10, 96, 3, [0], 13), (8, 3, 3, None, 0), (8, 700, 1, None, 0)])
def test_one_lookup_mdd(oracle, torch_cuda, length, count, min_distance, tolerance, masking):
    """Single segment MDD with a tolerance of 0 or 1 is a function of the observed word (mdd.cpp:43-80): the exact match, else
    the first barcode one mismatch away; the exact match comes before the quality masking, a masked read that is not a
    barcode goes through the scan, and so do bases outside ACGT and reads shorter than the barcode."""
    rng = np.random.default_rng(1300 + 10 * length + count)
    words = synthetic.barcode_set(rng, count, length, min_distance=min_distance)
    options = {"quality masking threshold": masking}
    if tolerance is not None:
        options["distance tolerance"] = tolerance
    spec = {"input segment cardinality": 1, "sample": synthetic.decoder_spec("mdd", [words], ["0::%d" % length], **options)}
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 16, 57 + length)
    code[100:140, length - 1] = 0                                      # reads one base short
    quality[100:140, length - 1] = 0
    code[200:220] = 15                                                 # all N
    code[300:320, 2] = 3                                               # an ambiguity code
    quality[400:500] = 2                                               # everything masked, exact matches among them
    for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
        s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)
    assert int((got[0]["status"] == 0).sum()) > (1 << 15)


@pytest.mark.parametrize("pairing,lengths,count,tolerance", [("zip", (8, 8), 48, [2, 2]), ("zip", (6, 9), 40, [2, 1]), ("product", (8, 8), 24, [1, 1]), ("product", (6, 7), 12, [1, 1])])
def test_mdd_tolerance_past_the_shannon_bound_in_a_compiled_job(oracle, torch_cuda, pairing, lengths, count, tolerance):
    """A compiled document handed in with a `distance tolerance` the compiler would have refused (metric.h:228): several
    words of a segment can then be within tolerance of one observation and mdd.cpp:50-80 returns the FIRST barcode in
    index order whose segments all fit, which one table lookup per segment cannot name.  Such a codec must take the scan
    kernels in both modes: e.g. segment 0 one mismatch from barcode a and two from barcode b, segment 1 on b exactly."""
    rng = np.random.default_rng(2900 + count + lengths[1])
    # words two mismatches apart at least (product) or three (pairs): the injected tolerances exceed (distance - 1) / 2
    first = synthetic.barcode_set(rng, count, lengths[0], 2 if pairing == "product" else 3)
    second = synthetic.barcode_set(rng, count, lengths[1], 2 if pairing == "product" else 3)
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("mdd", [first, second], ["0::%d" % lengths[0], "1::%d" % lengths[1]], pairing=pairing)}
    job = capi.compile_job(spec)
    job["sample"]["distance tolerance"] = list(tolerance)              # past the bound the compiler enforces
    L = sum(lengths)
    n = 1 << 15
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 91 + count)
    words, _ = synthetic._codec_arrays(job["sample"])
    letters = np.array([1, 2, 4, 8], dtype=np.uint8)
    # observations built to sit between barcodes: segment 0 of barcode a with one or two positions taken from barcode b,
    # segment 1 of barcode b
    N = words.shape[0]
    for i in range(4000):
        a, b = int(rng.integers(N)), int(rng.integers(N))
        row = words[a].copy()
        differing = np.nonzero(words[a, :lengths[0]] != words[b, :lengths[0]])[0]
        if differing.size:
            take = rng.choice(differing, size=min(differing.size, int(rng.integers(1, 4))), replace=False)
            row[take] = words[b, take]
        row[lengths[0]:L] = words[b, lengths[0]:L]
        code[2000 + i] = letters[row]
    for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
        s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)
    assert int((got[0]["status"] == 0).sum()) > n // 4


@pytest.mark.parametrize("lengths,count,min_distance,tolerance,masking", [
    ((8, 8), 96, 3, None, 0), ((10, 10), 196, 3, None, 20), ((6, 9), 48, 3, [0, 1], 0), ((10, 10), 24, 5, None, 13), ((5, 7), 30, 2, None, 0)])
def test_one_lookup_mdd_pairs(oracle, torch_cuda, lengths, count, min_distance, tolerance, masking):
    """Paired barcodes under MDD: a barcode is within both segments' tolerances only if each segment names it (one table
    lookup per segment, the nearest word up to two mismatches away), mdd.cpp:43-80; sets at distance 5 take a tolerance of 2."""
    rng = np.random.default_rng(1700 + lengths[0] + count)
    first, second = synthetic.barcode_set(rng, count, lengths[0], min_distance), synthetic.barcode_set(rng, count, lengths[1], min_distance)
    options = {"quality masking threshold": masking}
    if tolerance is not None:
        options["distance tolerance"] = tolerance
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("mdd", [first, second], ["0::%d" % lengths[0], "1::%d" % lengths[1]], pairing="zip", **options)}
    job = capi.compile_job(spec)
    L = sum(lengths)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 15, 23 + count)
    # the halves of different barcodes joined: each segment names another barcode
    words, _ = synthetic._codec_arrays(job["sample"])
    letters = np.array([1, 2, 4, 8], dtype=np.uint8)
    for i in range(200):
        code[1000 + i, :lengths[0]] = letters[words[i % count, :lengths[0]]]
        code[1000 + i, lengths[0]:L] = letters[words[(i + 1) % count, lengths[0]:L]]
    code[200:220] = 15
    code[300:320, 2] = 3
    quality[400:500] = 2
    # two mismatches in one segment
    code[600:700, 0] = letters[(np.searchsorted(letters, code[600:700, 0]) + 1) % 4]
    code[600:700, 1] = letters[(np.searchsorted(letters, code[600:700, 1]) + 1) % 4]
    for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
        s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)
    assert int((got[0]["status"] == 0).sum()) > (1 << 13)


def test_fast_mode_defers_only_a_small_fraction(oracle, torch_cuda):
    """The product form kernel finishes the clean reads itself; the worklist holds the ambiguous rest."""
    job, recipe = synthetic.config_job("C2")
    n = 1 << 18
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)
    confidence = got[0]["confidence"]
    ambiguous = float(((confidence < 0.8) | (got[0]["status"] == 2)).mean())
    assert ambiguous < 0.05


# ---- edge cases the reference's tests do not pin (SURVEY.md §4 gaps, §8a S1-S10) ------------------------------


def _small_job(algorithm="pamld", **extra):
    spec = {"input segment cardinality": 2,
            "sample": {"algorithm": algorithm, "transform": {"token": ["1::8"]},
                       "codec": {"@" + w: {"barcode": [w], "concentration": c} for w, c in
                                 (("AGGCAGAA", .18), ("CGTACTAG", .2), ("GGACTCCT", .22), ("TAAGGCGA", .23), ("TCCTGAGC", .17))}}}
    spec["sample"].update(extra)
    return jobspec.compile_job(spec)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 255, 256, 257, 1000])
def test_empty_and_ragged_batches(oracle, torch_cuda, n, mode):
    job, recipe = synthetic.config_job("C2")
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 99)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=mode)


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
@pytest.mark.parametrize("n", [1, 63, 65, 257, 1000])
def test_ragged_batches_of_the_streaming_kernels(oracle, torch_cuda, name, n):
    """The persistent kernels of round 3 (10 nt single index, two segment combinatorial sets, codecs of a few barcodes) on
    batches that end inside a 64 read tile, smaller than one tile, and of a single read."""
    job, recipe = synthetic.config_job(name)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 123 + n)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)


@pytest.mark.parametrize("mode", MODES)
def test_split_ranges_equal_one_call(oracle, torch_cuda, mode):
    job, recipe = synthetic.config_job("C5")
    code, quality, flag, offsets = synthetic.generate_rows(job, 5003, 5)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, pieces=7, mode=mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
def test_unknown_bases_zero_quality_and_short_observations(oracle, torch_cuda, algorithm, mode):
    job = _small_job(algorithm)
    rng = np.random.default_rng(11)
    n = 4096
    code = np.array([1, 2, 4, 8, 15, 0, 3], dtype=np.uint8)[rng.integers(0, 7, size=(n, 8))]
    quality = rng.integers(0, 45, size=(n, 8)).astype(np.uint8)
    quality[rng.random((n, 8)) < 0.1] = 0
    # all N, all Q0, and short observations: zero filled tail
    code[0] = 15; quality[0] = 2
    quality[1] = 0
    for i, keep in enumerate(range(8)):
        code[10 + i, keep:] = 0
        quality[10 + i, keep:] = 0
    code[20] = np.array([1, 4, 4, 2, 1, 4, 1, 0]); quality[20] = np.array([30] * 7 + [0])      # 7 of 8 observed, exact prefix
    quality[21] = 127
    quality[22] = 255       # BAM "missing" quality: masked to the table's 128 rows
    flag = (rng.random(n) < 0.2).astype(np.uint8)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, [0], mode=mode)


@pytest.mark.parametrize("mode", MODES)
def test_ties_keep_the_lowest_index(oracle, torch_cuda, mode):
    """strict '>' from zero.  Uniform priors, uniform qualities: equidistant barcodes tie exactly."""
    words = ["AAAAAAAA", "AAAACCCC", "CCCCAAAA", "CCCCCCCC", "GGGGGGGG", "TTTTTTTT"]
    spec = {"input segment cardinality": 1,
            "sample": {"algorithm": "pamld", "transform": {"token": ["0::8"]},
                       "codec": {"@" + w: {"barcode": [w]} for w in words}, "confidence threshold": 0.4}}
    job = jobspec.compile_job(spec)
    rng = np.random.default_rng(3)
    n = 2048
    letters = np.array([1, 2, 4, 8], dtype=np.uint8)
    code = letters[rng.integers(0, 4, size=(n, 8))]
    code[:64] = letters[np.array([0, 0, 1, 1, 0, 0, 1, 1])]            # AACCAACC: distance 4 to four barcodes
    quality = np.full((n, 8), 30, dtype=np.uint8)
    quality[64:] = np.array([37, 25, 12, 2], dtype=np.uint8)[rng.integers(0, 4, size=(n - 64, 8))]
    s, engine, got, _ = assert_parity(oracle, torch_cuda, job, code, quality, None, [0], mode=mode)
    assert set(got[0]["index"][:64]) <= {0, 1}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("priors", [(1, 1, 1, 1, 1, 1), (1, 2, 2, 1, 1, 1), (3, 1, 1, 3.0000000000000004, 1, 1), (1, 1, 1e-9, 1, 1, 1)])
def test_ties_and_priors(oracle, torch_cuda, mode, priors):
    """Equidistant barcodes under uniform qualities have the same conditional probability: the prior decides the argmax
    (pamld.cpp:66-79), the lowest index among equal products, and a prior one ulp larger wins.  With a confidence
    threshold of exactly 1/2 and two tied barcodes the posterior sits on the threshold (strict '>': low confidence)."""
    words = ["AAAAAAAA", "AAAACCCC", "CCCCAAAA", "CCCCCCCC", "GGGGGGGG", "TTTTTTTT"]
    spec = {"input segment cardinality": 1,
            "sample": {"algorithm": "pamld", "transform": {"token": ["0::8"]}, "noise": 0.0, "confidence threshold": 0.5,
                       "codec": {"@" + w: {"barcode": [w], "concentration": float(c)} for w, c in zip(words, priors)}}}
    job = jobspec.compile_job(spec)
    rng = np.random.default_rng(4)
    n = 2048
    letters = np.array([1, 2, 4, 8], dtype=np.uint8)
    code = letters[rng.integers(0, 4, size=(n, 8))]
    code[:64] = letters[np.array([0, 0, 1, 1, 0, 0, 1, 1])]            # AACCAACC: distance 4 to the first four barcodes
    code[64:128] = letters[np.array([0, 0, 0, 0, 0, 0, 1, 1])]         # AAAAAACC: distance 2 to the first two
    quality = np.full((n, 8), 30, dtype=np.uint8)
    quality[128:] = np.array([37, 25, 12, 2], dtype=np.uint8)[rng.integers(0, 4, size=(n - 128, 8))]
    assert_parity(oracle, torch_cuda, job, code, quality, None, [0], mode=mode)


@pytest.mark.parametrize("lengths,counts", [((6, 9), (12, 10)), ((10, 10), (24, 16)), ((4, 7, 5), (5, 6, 4)), ((16, 3), (7, 9)), ((1, 1), (4, 4)),
                                            ((7, 10, 4), (16, 14, 14)), ((12, 3, 9), (6, 9, 5))])
def test_combinatorial_sets_factorise(oracle, torch_cuda, lengths, counts):
    """Dual / triple index sets holding every combination go through the factorised kernel (pamld_cross.hpp) in
    PHQ_MODE_FAST: unequal segment lengths, segments that start off a 4 byte boundary, N calls, zero qualities,
    short observations, and uniform qualities (exact ties inside a segment, which must reach the literal kernel)."""
    rng = np.random.default_rng(sum(lengths) * 131 + sum(counts))
    segments = [synthetic.barcode_set(rng, c, l, min_distance=3 if l >= 6 else 1) for l, c in zip(lengths, counts)]
    tokens = ["%d::%d" % (k, l) for k, l in enumerate(lengths)]
    spec = {"input segment cardinality": len(lengths),
            "sample": synthetic.decoder_spec("pamld", segments, tokens, **{"confidence threshold": 0.9})}
    job = jobspec.compile_job(spec)
    n = 1 << 14
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 7)
    L = sum(lengths)
    noisy = rng.random((n, L)) < 0.02
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random((n, L)) < 0.02] = 0
    quality[: n // 8] = 30                                   # uniform qualities: exact ties between equidistant words
    for i in range(L):                                       # short observations: zero filled tail
        code[n // 2 + i, i:] = 0
        quality[n // 2 + i, i:] = 0
    quality[n // 2 + L] = 255
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 2       # most reads decode


@pytest.mark.parametrize("lengths,counts", [((10, 10), (24, 16)), ((6, 9), (12, 10)), ((8, 8), (40, 3)), ((16, 3), (7, 9))])
def test_combinatorial_sets_with_priors_that_differ(oracle, torch_cuda, lengths, counts):
    """Estimated priors (phq_adjust_prior) differ from barcode to barcode: the sum over a dual index codec no longer
    factorises.  The weighted scans (row and column of the winner exact, the rest bounded) and the weighted exact
    tier must reproduce the oracle: priors over two orders of magnitude, a few barcodes at zero."""
    rng = np.random.default_rng(sum(lengths) * 3 + sum(counts))
    segments = [synthetic.barcode_set(rng, c, l, min_distance=3 if l >= 6 else 1) for l, c in zip(lengths, counts)]
    tokens = ["%d::%d" % (k, l) for k, l in enumerate(lengths)]
    decoder = synthetic.decoder_spec("pamld", segments, tokens, **{"confidence threshold": 0.9})
    for k, record in enumerate(decoder["codec"].values()):
        record["concentration"] = 0 if k % 37 == 5 else float(10 ** rng.uniform(-1, 1))
    job = jobspec.compile_job({"input segment cardinality": len(lengths), "sample": decoder})
    n = 1 << 14
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 13)
    L = sum(lengths)
    noisy = rng.random((n, L)) < 0.02
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random((n, L)) < 0.02] = 0
    quality[: n // 8] = 30
    for i in range(L):
        code[n // 2 + i, i:] = 0
        quality[n // 2 + i, i:] = 0
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 2


@pytest.mark.parametrize("lengths,counts,spread,absent", [((10, 10), (40, 24), 1.0, 0), ((8, 8), (30, 20), 0.5, 0), ((6, 6), (20, 18), 1.0, 0),
                                                          ((9, 9), (24, 20), 3.0, 0), ((10, 10), (64, 48), 0.5, 0), ((10, 10), (40, 24), 1.0, 3),
                                                          ((8, 8), (24, 24), 0.5, 2)])
def test_combinatorial_neighbour_lists_with_priors(oracle, torch_cuda, lengths, counts, spread, absent):
    """Two segment combinatorial sets large enough for the neighbour list tier, priors spread over up to six decades: the
    tier weighs the winner's row and column by their own priors and bounds every other combination by the largest prior
    (pamld.cpp:66-86).  `absent`: every absent-th combination has prior zero (what a prior estimation pass leaves for a
    combination it saw nothing of) — a read naming one is left to the next tier.  The posterior stays within the fast
    mode's tolerance of the oracle's."""
    rng = np.random.default_rng(sum(lengths) * 5 + sum(counts))
    segments = [synthetic.barcode_set(rng, c, l, min_distance=3 if l >= 8 else 2) for l, c in zip(lengths, counts)]
    tokens = ["%d::%d" % (k, l) for k, l in enumerate(lengths)]
    decoder = synthetic.decoder_spec("pamld", segments, tokens, **{"confidence threshold": 0.95})
    for k, record in enumerate(decoder["codec"].values()):
        record["concentration"] = 0 if (absent and k % absent == 1) else float(10 ** rng.uniform(-spread, spread))
    job = jobspec.compile_job({"input segment cardinality": len(lengths), "sample": decoder})
    n = 1 << 15
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 17)
    L = sum(lengths)
    noisy = rng.random((n, L)) < 0.01
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random((n, L)) < 0.02] = 0
    quality[: n // 8] = 30
    quality[n // 8: n // 4] = 12
    if absent:          # reads ON the combinations whose prior is zero
        picks = rng.integers(0, counts[0] * counts[1], size=1 << 10)
        code[: 1 << 10] = np.array([[{"A": 1, "C": 2, "G": 4, "T": 8}[c] for c in segments[0][k // counts[1]] + segments[1][k % counts[1]]] for k in picks], dtype=np.uint8)
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 2


@pytest.mark.parametrize("lengths,counts,masking", [((6, 9), (12, 10), 0), ((10, 10), (24, 16), 15), ((4, 7, 5), (5, 6, 4), 0), ((16, 3), (7, 9), 10)])
def test_combinatorial_sets_factorise_under_mdd(oracle, torch_cuda, lengths, counts, masking):
    """The first barcode, in index order, whose every segment is within tolerance is (first fit of segment 1, ..):
    mdd_cross.hip scans the segments' distinct words.  Quality masking, N calls, short observations (the distance
    walks the OBSERVED length), tolerances at the Shannon bound and above what any word reaches."""
    rng = np.random.default_rng(sum(lengths) * 7 + sum(counts) + masking)
    segments = [synthetic.barcode_set(rng, c, l, min_distance=3 if l >= 6 else 1) for l, c in zip(lengths, counts)]
    tokens = ["%d::%d" % (k, l) for k, l in enumerate(lengths)]
    spec = {"input segment cardinality": len(lengths),
            "sample": synthetic.decoder_spec("mdd", segments, tokens, **{"quality masking threshold": masking})}
    job = jobspec.compile_job(spec)
    n = 1 << 14
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 11)
    L = sum(lengths)
    noisy = rng.random((n, L)) < 0.03
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random((n, L)) < 0.02] = 0
    for i in range(L):
        code[n // 2 + i, i:] = 0
        quality[n // 2 + i, i:] = 0
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 4           # masked qualities count as mismatches: fewer reads fit


def test_paired_sets_that_are_not_combinatorial_stay_exact(oracle, torch_cuda):
    """A zip paired dual index (word i with word i) is not a cross product: no factorised kernel, same results."""
    rng = np.random.default_rng(77)
    a, b = synthetic.barcode_set(rng, 24, 10), synthetic.barcode_set(rng, 24, 10)
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("pamld", [a, b], ["0::10", "1::10"], pairing="zip")}
    job = jobspec.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 13, 8)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)


@pytest.mark.parametrize("lengths,count,uniform", [((10, 10), 24, True), ((17,), 13, False), ((12, 12), 40, False), ((8, 8, 8, 8), 9, True), ((29,), 130, True),
                                                   # round 5: pairs of 11 to 16 nt segments take the paired tier by the first 10 nt of each segment
                                                   ((12, 12), 96, True), ((16, 16), 96, False), ((11, 14), 200, True), ((16, 16), 768, True)])
def test_long_barcodes_take_the_two_word_product_form(oracle, torch_cuda, lengths, count, uniform):
    """ACGT barcodes of 17 to 32 nt that are not combinatorial (zip paired segments, long single words) go through
    pamld_wide.hpp in PHQ_MODE_FAST: every group count 5..8, barcode counts off the block of 8, priors uniform or
    not, N calls, zero qualities, short observations, uniform qualities (ties)."""
    rng = np.random.default_rng(sum(lengths) * 17 + count)
    segments = [synthetic.barcode_set(rng, count, l, min_distance=3) for l in lengths]
    rng.shuffle(segments[-1])                                 # the pairing is arbitrary, not sorted against sorted
    tokens = ["%d::%d" % (k, l) for k, l in enumerate(lengths)]
    decoder = synthetic.decoder_spec("pamld", segments, tokens, pairing="zip", **{"confidence threshold": 0.9})
    if not uniform:
        for k, record in enumerate(decoder["codec"].values()):
            record["concentration"] = 1 + (k % 5)
    job = jobspec.compile_job({"input segment cardinality": len(lengths), "sample": decoder})
    n = 1 << 13
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 9)
    L = sum(lengths)
    noisy = rng.random((n, L)) < 0.02
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random((n, L)) < 0.02] = 0
    quality[: n // 8] = 30
    for i in range(L):
        code[n // 2 + i, i:] = 0
        quality[n // 2 + i, i:] = 0
    quality[n // 2 + L] = 255
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 2


def test_iupac_barcodes_take_the_uniform_branch(oracle, torch_cuda):
    """barcode letters outside ACGT always cost UNIFORM_BASE_QUALITY (the STRICT=false kernels)."""
    spec = {"input segment cardinality": 1,
            "cellular": [{"algorithm": "pamld", "transform": {"token": ["0::6"]},
                          "codec": {"@a": {"barcode": ["ACNTGA"]}, "@b": {"barcode": ["RCGTAA"]}, "@c": {"barcode": ["TTTTTT"]}}}]}
    job = jobspec.compile_job(spec)
    rng = np.random.default_rng(5)
    n = 4096
    code = np.array([1, 2, 4, 8, 15], dtype=np.uint8)[rng.integers(0, 5, size=(n, 6))]
    quality = rng.integers(0, 42, size=(n, 6)).astype(np.uint8)
    assert_parity(oracle, torch_cuda, job, code, quality, None, [0], mode=capi.MODE_LITERAL)
    assert_parity(oracle, torch_cuda, job, code, quality, None, [0], mode=capi.MODE_FAST)      # falls back to the literal kernel


def test_mdd_masking_tolerances_and_segments(oracle, torch_cuda):
    spec = {"input segment cardinality": 2,
            "sample": {"algorithm": "mdd", "transform": {"token": ["0::8", "1::10"]},
                       "quality masking threshold": 12, "distance tolerance": [1, 1],
                       "codec": {"@%d" % i: {"barcode": [a, b]} for i, (a, b) in enumerate(
                           [("AAAACCCC", "GGGGGTTTTT"), ("CCCCAAAA", "TTTTTGGGGG"), ("GTGTGTGT", "ACACACACAC"), ("TGCATGCA", "CATGCATGCA")])}}}
    job = jobspec.compile_job(spec)
    assert job["sample"]["shannon bound"] == [2, 3] and job["sample"]["distance tolerance"] == [1, 1]
    rng = np.random.default_rng(9)
    n = 1 << 14
    words = np.array([[ord(c) for c in "".join(r["barcode"])] for r in job["sample"]["codec"].values()], dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint8); lut[[65, 67, 71, 84]] = [1, 2, 4, 8]
    code = lut[words[rng.integers(0, 4, size=n)]]
    flip = rng.random((n, 18)) < 0.06
    code[flip] = np.array([1, 2, 4, 8, 15], dtype=np.uint8)[rng.integers(0, 5, size=int(flip.sum()))]
    quality = np.array([37, 25, 12, 11, 2], dtype=np.uint8)[rng.choice(5, size=(n, 18), p=[.7, .1, .1, .05, .05])]
    short = rng.random(n) < 0.05
    code[short, 14:18] = 0; quality[short, 14:18] = 0
    assert_parity(oracle, torch_cuda, job, code, quality, (rng.random(n) < 0.1).astype(np.uint8), [0])
    job["sample"]["quality masking threshold"] = 0
    assert_parity(oracle, torch_cuda, job, code, quality, None, [0])


def test_decoder_chain_sticky_qcfail_and_passthrough(oracle, torch_cuda):
    """QC fail set by one decoder feeds the pf counters of every later decoder."""
    spec = {"input segment cardinality": 2,
            "sample": _small_job()["sample"],
            "molecular": [{"algorithm": "naive", "transform": {"token": ["0::6"]}}],
            "cellular": [dict(_small_job("mdd")["sample"], **{"multiplexing classifier": False}),
                         dict(_small_job(**{"high quality distance threshold": 1, "high quality threshold": 20,
                                            "confidence threshold": 0.99})["sample"], **{"multiplexing classifier": False})]}
    for d in spec["cellular"]:
        d.pop("index", None)
    job = jobspec.compile_job({k: v for k, v in spec.items()})
    code, quality, flag, offsets = synthetic.generate_rows(job, 1 << 14, 21)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST)


# ---- properties at BASELINE.json's full size -----------------------------------------------------------------


def oracle_classify_threads(oracle, job, code, quality, flag, threads=16):
    """The oracle on `threads` host threads (a private engine each, merged by its own collect): (records, status words,
    the engine that holds the merged counters)."""
    from concurrent.futures import ThreadPoolExecutor
    n = code.shape[0]
    bounds = np.linspace(0, n, threads + 1).astype(np.int64)
    engines = [oracle.OracleJob(job) for _ in range(threads)]
    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(lambda k: engines[k].classify(code[bounds[k]:bounds[k + 1]], quality[bounds[k]:bounds[k + 1]], flag[bounds[k]:bounds[k + 1]]), range(threads)))
    for other in engines[1:]:
        engines[0].collect(other)
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), engines[0]


# BASELINE.json's sizes per GPU: C2 and C3 100 M reads on one, C4 500 M over four, C5 1 B over eight (125 M each)
FULL_SIZE = [("C2", 1 << 18, 382), ("C3", 1 << 16, 1526), ("C4", 1 << 17, 954), ("C5", 1 << 17, 954)]


@pytest.mark.parametrize("config,pool_n,tiles", FULL_SIZE)
@pytest.mark.parametrize("mode", MODES)
def test_full_size_is_a_tiling_of_the_oracle_checked_pool(oracle, torch_cuda, mode, config, pool_n, tiles):
    """BASELINE.json's full size per GPU (100 M reads of C2 and C3, 125 M of C4 and C5) built by tiling a pool the oracle can
    decode in seconds: every counter of every decoder must be exactly `tiles` times the pool's, and every result record
    of every decoder equal to the pool's record.  Literal mode walks every barcode of every read in fp64: it takes the
    single index shape only (C3's 9 216 barcodes at 100 M reads are minutes of GPU time)."""
    if mode == capi.MODE_LITERAL and config != "C2":
        pytest.skip("literal mode at full size: C2 only")
    torch = torch_cuda
    job, recipe = synthetic.config_job(config)
    code, quality, flag, offsets = synthetic.generate_rows(job, pool_n, recipe["seed"])
    expected, expected_word, engine = oracle_classify_threads(oracle, job, code, quality, flag)
    s = DecoderSet(job, device=0)
    s.set_mode(mode)
    n = pool_n * tiles

    def resident(rows):
        pool_planes = torch.from_numpy(s.planes_from_rows(rows, offsets)).cuda()
        parts = [pool_planes[pool_n * d["plane_prefix"]: pool_n * (d["plane_prefix"] + d["plane_stride"])].view(pool_n, d["plane_stride"]).repeat(tiles, 1).view(-1)
                 for d in s.decoders if d["plane_stride"]]
        return torch.cat(parts) if len(parts) > 1 else parts[0]

    base, qual = resident(code), resident(quality)
    flags = torch.from_numpy(flag).cuda().repeat(tiles)
    result = torch.zeros(s.n_result * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    word = torch.zeros(n, dtype=torch.int32, device="cuda")
    s.classify_device(n, 0, n, base, qual, flags, result, word, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    del base, qual
    words = s.stats_export()
    assert int(words[0] + words[1]) == n
    count, pf_count = engine.outgoing()
    assert (int(words[0] + words[1]), int(words[1])) == (count * tiles, pf_count * tiles)
    for k in range(s.n_decoder):
        mine, theirs = s.stats_options(words, k), engine.options(k)
        for name in INTEGER_COUNTERS:
            assert np.array_equal(mine[name], theirs[name] * tiles), (k, name)
    # idempotence across tiles: every tile's records equal the first tile's, and the first equals the oracle's
    for slot in range(s.n_result):
        records = result[slot * n * RESULT_DTYPE.itemsize:(slot + 1) * n * RESULT_DTYPE.itemsize].view(tiles, pool_n * RESULT_DTYPE.itemsize)
        assert bool((records == records[0]).all()), slot
        first = records[0].cpu().numpy().view(RESULT_DTYPE)
        for name in INTEGER_FIELDS:
            assert np.array_equal(first[name], expected[:, slot][name]), (slot, name)
        assert np.abs(first["confidence"] - expected[:, slot]["confidence"]).max() < POSTERIOR_DESIGN_TOLERANCE[mode]
    assert bool((word.view(tiles, pool_n) == word.view(tiles, pool_n)[0]).all())
    assert np.array_equal(word[:pool_n].cpu().numpy().view(np.uint32), expected_word)
    s.close()


@pytest.mark.parametrize("mode", MODES)
def test_classify_host_equals_classify_device(oracle, torch_cuda, mode):
    job, recipe = synthetic.config_job("C5")
    n = (1 << 22) + 12345                                              # more than one pipeline chunk
    pool = 1 << 15
    code, quality, flag, offsets = synthetic.generate_rows(job, pool, 77)
    engine = oracle.OracleJob(job)
    expected, expected_word = engine.classify(code, quality, flag)
    reps = -(-n // pool)
    code = np.tile(code, (reps, 1))[:n]; quality = np.tile(quality, (reps, 1))[:n]; flag = np.tile(flag, reps)[:n]
    s = DecoderSet(job, device=0)
    s.set_mode(mode)
    result, word = s.classify_host(n, s.planes_from_rows(code, offsets), s.planes_from_rows(quality, offsets), flag)
    index = np.arange(n) % pool
    for slot in range(s.n_result):
        for name in INTEGER_FIELDS:
            assert np.array_equal(result[slot][name], expected[index, slot][name]), (slot, name)
    assert np.array_equal(word, expected_word[index])
    words = s.stats_export()
    assert int(words[0] + words[1]) == n


def test_counter_allreduce_over_rccl_single_rank(oracle, torch_cuda):
    """The counter buffer is exposed as a device tensor; a one rank RCCL all-reduce must leave it unchanged."""
    import os
    import torch.distributed as dist
    torch = torch_cuda
    job, recipe = synthetic.config_job("C2")
    code, quality, flag, offsets = synthetic.generate_rows(job, 4096, 1)
    s, got, word, words = run_gpu(torch, job, code, quality, flag, offsets)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        tensor = s.stats_tensor()
        assert tensor.dtype == torch.int64 and tensor.numel() == s.stats_words
        dist.all_reduce(tensor)
        torch.cuda.synchronize()
        assert np.array_equal(s.stats_export(), words)
    finally:
        dist.destroy_process_group()


# ---- the steps either side of the path: FASTQ in, SAM records out ---------------------


@pytest.mark.parametrize("batch_capacity,compressed", [(1 << 16, False), (100, False), (64, True)])
def test_bdggg_fastq_to_sam_is_the_golden_file(torch_cuda, tmp_path, batch_capacity, compressed):
    """The reference's own compiled job (valid/compile_annotated.out, unchanged) and its three FASTQ files through
    phq_transcode_fastq_to_sam: valid/annotated.out byte for byte, header lines included.  (@PG names the program
    and its command line; the reference's own test drops it before comparing, test/BDGGG/function.sh:29.)"""
    import gzip
    import os
    import shutil
    inputs = []
    for k in (1, 2, 3):
        source = os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k)
        if compressed:
            target = os.path.join(tmp_path, "s0%d.fastq.gz" % k)
            with open(source, "rb") as a, gzip.open(target, "wb") as b:
                shutil.copyfileobj(a, b)
            source = target
        inputs.append(source)
    s = DecoderSet(json.dumps(bdggg.compiled_reference_job()), device=0)
    output = os.path.join(tmp_path, "out.sam")
    incoming = s.transcode_fastq_to_sam(inputs, output, batch_capacity)
    assert incoming == (250, 243)
    golden = open(os.path.join(bdggg.HERE, "annotated.out.sam")).readlines()
    mine = [line for line in open(output) if not line.startswith("@PG")]
    assert len(mine) == len(golden) == 496 + 7
    for a, b in zip(mine, golden):
        assert a == b
    # and the report of the same run, against valid/annotated.err
    report = json.loads(s.report(s.stats_export(), incoming))
    theirs = bdggg.golden_report()
    assert report["incoming"] == theirs["incoming"]
    assert report["outgoing"]["count"] == theirs["outgoing"]["count"] and report["outgoing"]["pf count"] == theirs["outgoing"]["pf count"]
    for mine_block, their_block in ((report["sample"], theirs["sample"]), (report["cellular"][0], theirs["cellular"][0]), (report["molecular"][0], theirs["molecular"][0])):
        for key in ("count", "pf count", "classified count", "pf classified count"):
            assert mine_block[key] == their_block[key]
        for m, t in zip(mine_block.get("classified", []), their_block.get("classified", [])):
            assert (m["count"], m["pf count"], m.get("low confidence count"), m.get("low conditional confidence count")) == \
                   (t["count"], t["pf count"], t.get("low confidence count"), t.get("low conditional confidence count"))
            assert abs(m["average confidence"] - t["average confidence"]) < 1e-12


def test_large_batches_write_the_same_records_in_the_same_order(torch_cuda, tmp_path):
    """Staging and record formatting split a large batch over host threads, and the next batch is read while the
    current one is written: the SAM text must not depend on the batch size (small batches take the serial path)."""
    import os
    rng = np.random.default_rng(21)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    spec["min input length"] = [0, 6]
    job = jobspec.compile_job(spec)
    n = 70001
    code, quality, flag, _ = synthetic.generate_rows(job, n, 12)
    letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 30))]
    paths = [os.path.join(tmp_path, "s%d.fastq" % k) for k in (0, 1)]
    with open(paths[0], "wb") as a, open(paths[1], "wb") as b:
        for i in range(n):
            filtered = b"Y" if flag[i] else b"N"
            keep = 8 if i % 97 else 5                        # a few index reads too short for `min input length`
            a.write(b"@r%d 1:%s:0:1\n%s\n+\n%s\n" % (i, filtered, template[i].tobytes(), b"F" * 30))
            b.write(b"@r%d 2:%s:0:1\n%s\n+\n%s\n" % (i, filtered, letters[code[i, :keep]].tobytes(), (quality[i, :keep] + 33).astype(np.uint8).tobytes()))
    outputs = []
    for batch_capacity in (4096, 1 << 17):
        s = DecoderSet(job, device=0)
        output = os.path.join(tmp_path, "out_%d.sam" % batch_capacity)
        incoming = s.transcode_fastq_to_sam(paths, output, batch_capacity)
        assert incoming[0] == n
        outputs.append((open(output, "rb").read(), s.stats_export()))
    assert outputs[0][0] == outputs[1][0]
    assert np.array_equal(outputs[0][1], outputs[1][1])
    records = [line for line in outputs[0][0].split(b"\n") if not line.startswith(b"@")]
    kept = n - len(range(0, n, 97))
    assert len(records) - 1 == 2 * kept
    assert records[0].startswith(b"r1\t") and records[-2].startswith(b"r%d\t" % (n - 1))


def _bdggg_job_with_io(tmp_path, **changes):
    """BDGGG_annotated.json as a user writes it (imports resolved, nothing compiled), reading the committed FASTQ files
    and writing under tmp_path."""
    job = bdggg.raw_job()
    job["base input url"] = bdggg.HERE
    job["base output url"] = str(tmp_path)
    job.update(changes)
    return job


def _golden_sam_records():
    """valid/annotated.out: (header lines, [(name, flag, sequence, quality, {tag: value}) per record])."""
    header, records = [], []
    for line in open(os.path.join(bdggg.HERE, "annotated.out.sam")):
        if line.startswith("@"):
            header.append(line)
            continue
        field = line.rstrip("\n").split("\t")
        tags = {t[:2]: t[5:] for t in field[11:]}
        records.append((field[0], int(field[1]), field[9], field[10], tags))
    return header, records


def test_job_directives_user_level_config_to_the_golden_file(torch_cuda, tmp_path):
    """phq_transcode_job: the user level configuration unchanged but for the directories — compiled by the library,
    `input` under `base input url`, a single `output` path under `base output url` repeated for both output segments —
    gives valid/annotated.out byte for byte."""
    job = _bdggg_job_with_io(tmp_path, output=["out.sam"])
    s = DecoderSet(json.dumps(job), device=0)
    assert s.transcode_job(100) == (250, 243)
    golden = open(os.path.join(bdggg.HERE, "annotated.out.sam")).readlines()
    mine = [line for line in open(os.path.join(tmp_path, "out.sam")) if not line.startswith("@PG")]
    assert mine == golden


def test_copies_of_the_set_behind_one_pull_loop(torch_cuda, tmp_path):
    """phq_transcode_job_copies (reference transcode.cpp:1769-1775: N decoder copies behind one pull loop; 1816-1818: their
    counters merged at the end): the BDGGG job in batches of 16 reads dealt to the set and two copies of it — on this one
    GPU pool all three on device 0, on a node one per device — writes valid/annotated.out byte for byte, and after the run the
    set's counters are the whole run's (word for word a single set's), the copies' zero."""
    job = _bdggg_job_with_io(tmp_path, output=["out.sam"])
    devices = list(range(min(3, torch_cuda.cuda.device_count()))) or [0]
    s = DecoderSet(json.dumps(job), device=0)
    copies = [DecoderSet(json.dumps(job), device=devices[(k + 1) % len(devices)]) for k in range(2)]
    for each in [s] + copies:
        each.stats_reset()
    assert s.transcode_job_copies(copies, 16) == (250, 243)
    golden = open(os.path.join(bdggg.HERE, "annotated.out.sam")).readlines()
    mine = [line for line in open(os.path.join(tmp_path, "out.sam")) if not line.startswith("@PG")]
    assert mine == golden
    merged = s.stats_export()
    for copy in copies:
        assert int(copy.stats_export().sum()) == 0
    alone = DecoderSet(json.dumps(job), device=0)
    alone.stats_reset()
    assert alone.transcode_job(16) == (250, 243)
    assert np.array_equal(merged, alone.stats_export())
    for each in [alone, s] + copies:
        each.close()


def test_job_directives_interleaved_input_feed(torch_cuda, tmp_path):
    """A path stated three times in `input` is ONE feed whose consecutive records are the three segments of a read
    (transcode.cpp:693-701): the same output as from the three separate files; a feed that ends inside a read and
    segments out of sync are sequence errors."""
    sources = [open(os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k)).read().split("\n") for k in (1, 2, 3)]
    n = min(len(lines) // 4 for lines in sources)
    path = os.path.join(tmp_path, "interleaved.fastq")
    with open(path, "w") as out:
        for i in range(n):
            for lines in sources:
                out.write("\n".join(lines[4 * i:4 * i + 4]) + "\n")
    job = _bdggg_job_with_io(tmp_path, input=[path, path, path], output=["out.sam?format=sam"])
    s = DecoderSet(json.dumps(job), device=0)
    assert s.transcode_job(64) == (250, 243)
    golden = open(os.path.join(bdggg.HERE, "annotated.out.sam")).readlines()
    mine = [line for line in open(os.path.join(tmp_path, "out.sam")) if not line.startswith("@PG")]
    assert mine == golden
    # one record short: the feed ends inside a read
    with open(path, "w") as out:
        for i in range(10):
            for k, lines in enumerate(sources):
                if i == 9 and k == 2: break
                out.write("\n".join(lines[4 * i:4 * i + 4]) + "\n")
    with pytest.raises(PheniqsError) as caught:
        DecoderSet(json.dumps(job), device=0).transcode_job(64)
    assert "ends inside a read" in str(caught.value)
    # the second segment of read 5 taken from read 6: identifiers no longer agree
    with open(path, "w") as out:
        for i in range(10):
            for k, lines in enumerate(sources):
                at = 4 * (i + 1 if (i == 5 and k == 1) else i)
                out.write("\n".join(lines[at:at + 4]) + "\n")
    with pytest.raises(PheniqsError) as caught:
        DecoderSet(json.dumps(job), device=0).transcode_job(64)
    assert "out of sync" in str(caught.value)


def test_job_directives_split_fastq_outputs_by_channel(torch_cuda, tmp_path):
    """Multiplexer::push (multiplex.h:198-240): every sample barcode is a channel with its own files.  Two gzip FASTQ
    files per barcode (one per output segment), the undetermined reads interleaved into one plain FASTQ file, one
    barcode as SAM, one barcode without QC failed reads.  Expected records are derived from valid/annotated.out: the
    read group names the channel, the FASTQ comment is `segment:Y|N:0:BC` (fastq.h:176-194)."""
    import gzip
    job = _bdggg_job_with_io(tmp_path)
    codec = job["sample"]["codec"]
    for key in codec:
        codec[key]["output"] = ["%s_s1.fastq.gz" % key[1:], "%s_s2.fastq.gz" % key[1:]]
    codec["@CGTACTAG"]["output"] = ["CGTACTAG.sam"]
    codec["@GGACTCCT"]["filter outgoing qc fail"] = True
    job["sample"]["undetermined"] = {"output": ["undetermined.fq"]}
    s = DecoderSet(json.dumps(job), device=0)
    assert s.transcode_job(77) == (250, 243)

    header, records = _golden_sam_records()
    def fastq(record):
        name, flag, sequence, quality, tags = record
        return "@%s %d:%s:0:%s\n%s\n+\n%s\n" % (name, 1 if flag & 64 else 2, "Y" if flag & 512 else "N", tags.get("BC", ""), sequence, quality)
    by_group = {}
    for record in records:
        by_group.setdefault(record[4]["RG"].split(":")[-1], []).append(record)
    assert set(by_group) == {"undetermined", "AGGCAGAA", "CGTACTAG", "GGACTCCT", "TAAGGCGA", "TCCTGAGC"}
    for barcode in ("AGGCAGAA", "GGACTCCT", "TAAGGCGA", "TCCTGAGC"):
        for segment, bit in ((1, 64), (2, 128)):
            expected = "".join(fastq(r) for r in by_group[barcode] if r[1] & bit and not (barcode == "GGACTCCT" and r[1] & 512))
            assert gzip.open(os.path.join(tmp_path, "%s_s%d.fastq.gz" % (barcode, segment)), "rt").read() == expected, (barcode, segment)
    assert any(r[1] & 512 for r in by_group["GGACTCCT"])                      # the filter had something to drop
    assert open(os.path.join(tmp_path, "undetermined.fq")).read() == "".join(fastq(r) for r in by_group["undetermined"])
    sam = [line for line in open(os.path.join(tmp_path, "CGTACTAG.sam")) if not line.startswith("@PG")]
    golden = [line for line in open(os.path.join(bdggg.HERE, "annotated.out.sam")) if line.startswith("@") or "RG:Z:BDGGG:1:CGTACTAG" in line]
    assert sam == golden
    # counters do not depend on where the reads are written
    report = json.loads(s.report(s.stats_export(), (250, 243)))
    assert report["outgoing"] == bdggg.golden_report()["outgoing"]


def test_job_directives_errors(torch_cuda, tmp_path):
    """Configuration errors of compile_output (transcode.cpp:1337-1345,1446-1460,1383-1393) and formats that need htslib."""
    def run(**changes):
        job = _bdggg_job_with_io(tmp_path, **changes)
        DecoderSet(json.dumps(job), device=0).transcode_job(64)
    for changes, text in (({"output": ["a.sam", "b.sam", "c.sam"]}, "incorrect number of output URLs in channel"),
                          ({"output": ["/dev/stdin"]}, "output stream can not be set to standard input"),
                          ({"output": ["/dev/stderr"]}, "output stream can not be set to standard error"),
                          ({"output": ["out.cram"]}, "htslib"),
                          ({"output": ["out.fastq.bz2"]}, "not supported"),
                          ({"input": ["a.bam", "BDGGG_s02.fastq", "BDGGG_s03.fastq"]}, "only FASTQ input"),
                          ({"input": ["missing.fastq", "BDGGG_s02.fastq", "BDGGG_s03.fastq"]}, "could not open")):
        with pytest.raises(PheniqsError) as caught:
            run(**changes)
        assert text in str(caught.value), (changes, str(caught.value))
    job = _bdggg_job_with_io(tmp_path)
    job["sample"]["codec"]["@AGGCAGAA"]["output"] = ["x.fastq", "x.fastq"]
    job["sample"]["codec"]["@CGTACTAG"]["output"] = ["x.fastq", "y.fastq"]
    job["output"] = ["rest.sam"]
    with pytest.raises(PheniqsError) as caught:
        DecoderSet(json.dumps(job), device=0).transcode_job(64)
    assert "inconsistent resolution" in str(caught.value)


@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
def test_large_codec_counters_in_a_chain_with_sticky_qcfail(oracle, torch_cuda, algorithm):
    """A codec too large for the workgroup histogram (its classified reads are counted from the result records by
    count_results_kernel) in front of and behind small codecs: pass filter / QC fail halves follow the sticky flag
    (SURVEY S3), the high quality distance filter and the incoming flags."""
    rng = np.random.default_rng(5)
    big = [synthetic.barcode_set(rng, 24, 8), synthetic.barcode_set(rng, 16, 8)]           # 384 combinations
    small = synthetic.barcode_set(rng, 24, 8)
    options = {"high quality distance threshold": 2, "high quality threshold": 20} if algorithm == "pamld" else {}
    spec = {"input segment cardinality": 4,
            "sample": synthetic.decoder_spec(algorithm, [small], ["0::8"], **options),
            "molecular": [{"algorithm": "naive", "transform": {"token": ["3::6"]}}],
            "cellular": [synthetic.decoder_spec(algorithm, big, ["1::8", "2::8"], **options),
                         synthetic.decoder_spec(algorithm, [small], ["0::8"])]}
    job = jobspec.compile_job(spec)
    n = 30000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 4)
    flag = (rng.random(n) < 0.3).astype(np.uint8)
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, pieces=3)
    options_big = s.stats_options(words, 2)
    assert int(options_big["count"][1:].sum()) > n // 2 and int(options_big["pf_count"][1:].sum()) < int(options_big["count"][1:].sum())


def test_high_quality_threshold_beyond_the_byte_parallel_compare(oracle, torch_cuda):
    """Thresholds above 128 are outside the product form kernels' byte parallel quality compare: those decoders
    take the literal kernel, with the same results."""
    job = _small_job("pamld", **{"high quality threshold": 200, "high quality distance threshold": 1})
    code, quality, flag, offsets = synthetic.generate_rows(job, 4096, 2)
    quality[:64] = 255
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)


@pytest.mark.parametrize("length,count,shared", [(11, 96, False), (12, 96, False), (13, 200, False), (16, 384, False), (20, 96, False), (12, 96, True), (16, 300, True)])
def test_single_index_sets_of_11_to_20_nt_take_the_paired_tier_in_two_parts(oracle, torch_cuda, length, count, shared):
    """A single index past the streaming kernel's 10 nt is handed to the paired tier as two parts of its one segment (8 nt and
    the rest): both modes equal the oracle, `phq_profile_chain` names pamld_pair_kernel first.  `shared`: many barcodes begin
    with the same 8 nt or end with the same words — such words name no one (PAIR_ANY_BARCODE) and the other part decides; reads
    are also put ON barcodes whose both parts are shared, which no table names."""
    rng = np.random.default_rng(length * 100 + count + (7 if shared else 0))
    words = synthetic.barcode_set(rng, count, length, min_distance=4)
    if shared:
        # every fourth barcode takes the first 8 nt of its neighbour, every fifth the last part of its neighbour; some both
        mixed = []
        for k, word in enumerate(words):
            other = words[(k + 1) % count]
            head = other[:8] if k % 4 == 0 else word[:8]
            tail = other[8:] if k % 5 == 0 else word[8:]
            mixed.append(head + tail)
        words = sorted(set(mixed))
    job = jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words], ["0::%d" % length])})
    n = 40000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 9)
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    s.set_mode(capi.MODE_FAST)
    s.profile_enable(True)
    run_gpu(torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST, device_set=s)
    chain = [span["kernel"] for span in s.profile_chain()]
    if not shared:
        assert chain[0] == "pamld_pair_kernel", chain
    assert int((got[0]["status"] == 0).sum()) > n // 2
    s.close()


@pytest.mark.parametrize("length,count,masking,shared", [(11, 60, 0, False), (12, 96, 25, False), (16, 200, 30, False), (20, 96, 0, False), (12, 96, 25, True)])
def test_single_index_sets_of_11_to_20_nt_under_mdd_take_the_paired_lookup_in_two_parts(oracle, torch_cuda, length, count, masking, shared):
    """The same two parts under MDD (mdd_near.hpp): either part's word names the candidate, the tolerance is the whole barcode's,
    only a read that IS classified is finished by the lookup and the scan takes the rest.  With quality masking the reference
    settles exact matches first (mdd.cpp:43-46), which the lookup can do only for barcodes a part of their own names: a set
    where some barcode shares both its part words with others (`shared`) keeps the old chain.  Reads ON barcodes with low
    qualities (masked positions) are in the batch."""
    rng = np.random.default_rng(length * 131 + count + masking)
    words = synthetic.barcode_set(rng, count, length, min_distance=5 if length >= 12 else 3)
    if shared:
        first = 8 if length <= 16 else length - 10
        words = sorted(set(words + [words[0][:first] + words[1][first:], words[1][:first] + words[0][first:]]))      # four barcodes over two words a part
    extra = {"quality masking threshold": masking} if masking else {}
    if length >= 16:
        extra["distance tolerance"] = [2]                  # (words this long are far apart: their Shannon bound is past the tables' reach of two)
    job = jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec("mdd", [words], ["0::%d" % length], **extra)})
    n = 40000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 10)
    letters = {"A": 1, "C": 2, "G": 4, "T": 8}
    for k in range(2000):                                   # reads ON a barcode, some positions under any masking threshold
        code[k] = [letters[c] for c in words[k % len(words)]]
        quality[k] = np.where(rng.random(length) < 0.3, 5, 37)
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    s.set_mode(capi.MODE_FAST)
    s.profile_enable(True)
    run_gpu(torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST, device_set=s)
    chain = [span["kernel"] for span in s.profile_chain()]
    assert (chain[0] == "mdd_lookup_kernel (paired)") == (not shared), chain
    assert int((got[0]["status"] == 0).sum()) > n // 2
    s.close()


@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
@pytest.mark.parametrize("lengths,count,pairing", [((33,), 24, "zip"), ((40,), 96, "zip"), ((20, 20), 96, "zip"), ((20, 20), 6, "product"),
                                                   ((12, 12, 12), 4, "product"), ((64,), 12, "zip"), ((100, 28), 8, "zip"), ((128,), 5, "zip")])
def test_barcodes_of_33_to_128_nt_take_the_general_kernels(oracle, torch_cuda, algorithm, lengths, count, pairing):
    """The reference loops over any barcode length and any segment array (barcode.h:147-161, sequence.h:90-98); every tier here
    stops at 32 nt.  Past them a job is no longer refused: pamld_literal_long.hip walks every barcode for every
    read with the literal kernel's arithmetic and argmax rules, mdd_kernel runs over 8 or 16 packed words — in both modes, with
    bases outside ACGT, short reads (zero filled), a barcode letter outside ACGT (S7), quality masking and reads ON
    barcodes at low qualities in the batch."""
    rng = np.random.default_rng(sum(lengths) * 977 + count)
    segments = [synthetic.barcode_set(rng, count, length, min_distance=max(5, length // 3 + 1)) for length in lengths]
    total = sum(lengths)
    if algorithm == "pamld" and total == 40 and len(lengths) == 1:
        segments[0][3] = segments[0][3][:7] + "N" + segments[0][3][8:]          # not every barcode letter is one of A, C, G, T
        segments[0] = sorted(segments[0])
    tokens, start = [], 0
    for length in lengths:
        tokens.append("0:%d:%d" % (start, start + length))
        start += length
    extra = {}
    if algorithm == "mdd":
        extra = {"distance tolerance": [max(2, length // 6) for length in lengths], "quality masking threshold": 10 if total % 2 == 0 else 0}       # (2.4 % of the calls are errors)
    job = jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec(algorithm, segments, tokens, pairing=pairing, **extra)})
    n = 6000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 21)
    code[rng.random((n, total)) < 0.004] = 15
    short = rng.integers(0, n, 40)
    for k, i in enumerate(short):                          # what Rule::apply leaves of a read that ends early
        keep = total - 1 - (k % 9)
        code[i, keep:] = 0
        quality[i, keep:] = 0
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    s.set_mode(capi.MODE_FAST)
    s.profile_enable(True)
    run_gpu(torch_cuda, job, code, quality, flag, offsets, mode=capi.MODE_FAST, device_set=s)
    chain = [span["kernel"] for span in s.profile_chain()]
    assert chain == ["pamld_literal_long_kernel (every read)" if algorithm == "pamld" else "mdd_kernel (every read)"], chain
    assert int((got[0]["status"] == 0).sum()) > n // 2
    s.close()


def test_barcodes_past_128_nt_are_a_configuration_error():
    """DEV_MAX_NUCLEOTIDE: 16 packed words of a barcode; the error is the reference's ErrorCode for a bad document (error.h:32-44)"""
    rng = np.random.default_rng(4)
    words = synthetic.barcode_set(rng, 4, 129, min_distance=5)
    with pytest.raises(PheniqsError) as failure:
        DecoderSet(jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words], ["0::129"])}), device=0)
    assert failure.value.code == 3 and "128" in str(failure.value)          # ErrorCode::CONFIGURATION_ERROR


@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
@pytest.mark.parametrize("length,count", [(1, 2), (2, 3), (4, 16), (5, 7), (8, 1), (8, 512), (8, 520), (8, 1024), (8, 1030), (12, 33),
                                          (16, 256), (16, 260), (16, 512), (16, 515), (32, 5),
                                          # round 5: past 512 packed words the product form stays the first tier (it was the literal
                                          # kernel for every read): 768 / 1 536 / 4 096 barcodes, and the first size past the new limit
                                          (8, 1536), (10, 768), (12, 1536), (16, 4096), (16, 4100), (20, 1536), (20, 2050)])
def test_codec_size_extremes(oracle, torch_cuda, algorithm, length, count):
    """The smallest and the largest codecs each kernel family takes, and the first sizes past them (which fall back
    to the next family): one barcode, one nucleotide, 1 024 barcodes of 8 nt in 512 packed words, 512 of 16 nt; 512 /
    256 were the limits of the first version and stay as cases."""
    rng = np.random.default_rng(length * 1000 + count)
    words = synthetic.barcode_set(rng, count, length, min_distance=1 if length < 6 or count > 200 else 3)
    job = jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec(algorithm, [words], ["0::%d" % length])})
    n = 4096
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 3)
    code[rng.random((n, length)) < 0.01] = 15
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)


@pytest.mark.parametrize("priors", [False, True])
@pytest.mark.parametrize("length,count", [(8, 600), (10, 768), (12, 1024), (16, 768), (16, 1024), (20, 768), (24, 1024), (32, 600), (20, 1025),
                                          (8, 1536), (12, 1536), (16, 2048), (16, 2049), (20, 1536), (32, 2048), (24, 2049)])
def test_the_exact_tiers_hold_up_to_2048_barcodes(oracle, torch_cuda, length, count, priors):
    """Round 6: the fp64 tiers between the product forms and the literal kernel — pamld_exact_lanes (words of up to 16 nt) and
    pamld_wide_exact (17 to 32 nt) — keep their word lists and list masks for 2 048 barcodes (512 until now: a
    larger set handed whatever its product form could not certify straight to the literal kernel, half the step of 768 x 8 nt), and the
    lookup tiers in front of them (pamld_near, pamld_pair) take sets of up to 2 048 (1 024).  Sets of 600 to 2 049 barcodes with uniform
    priors and priors that differ, noise reads and reads without a called base among them, both modes against the oracle; the plan
    names the tier up to its size and not past it."""
    rng = np.random.default_rng(9000 + 10 * length + count + (1 if priors else 0))
    words = synthetic.barcode_set(rng, count, length, min_distance=1 if length < 10 else 3)
    spec = {"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words], ["0::%d" % length], noise=0.05)}
    if priors:
        for record in spec["sample"]["codec"].values():
            record["concentration"] = float(10 ** rng.uniform(-1.5, 1.5))
    job = capi.compile_job(spec)
    n = 30000 if count <= 1024 else 12000               # (the oracle's walk over 2 048 barcodes is what the case costs)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 5 + length)
    code[rng.random((n, length)) < 0.01] = 15
    s, engine, got, words_out = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    kernels = [c["kernel"] for c in s.plan_chain(100_000_000, capi.MODE_FAST)]
    tier = "pamld_exact_lanes_kernel" if length <= 16 else "pamld_wide_exact_kernel"
    assert (tier in kernels) == (count <= 2048), kernels


def test_many_decoders_and_many_segments(oracle, torch_cuda):
    """The widest job the library takes: 16 decoders (DEV_MAX_DECODER) of every algorithm in one chain — the sticky QC
    fail flag and the channel index travel through all of them — among them barcodes of 5 and 8 segments
    (DEV_MAX_SEGMENT) under MDD and under PAMLD."""
    rng = np.random.default_rng(16)
    def words(count, length):
        return synthetic.barcode_set(rng, count, length, min_distance=1 if length < 5 else 2)
    five = [words(6, 3) for _ in range(5)]
    eight = [words(5, 2) for _ in range(8)]
    for w in five[1:] + eight[1:]:
        rng.shuffle(w)
    cellular = [synthetic.decoder_spec("mdd", five, ["0:%d:%d" % (3 * k, 3 * k + 3) for k in range(5)], pairing="zip"),
                synthetic.decoder_spec("pamld", eight, ["1:%d:%d" % (2 * k, 2 * k + 2) for k in range(8)], pairing="zip"),
                synthetic.decoder_spec("pamld", [words(40, 8)], ["0:15:23"], **{"high quality distance threshold": 1, "high quality threshold": 20}),
                synthetic.decoder_spec("mdd", [words(30, 6)], ["1:16:22"], **{"quality masking threshold": 12}),
                synthetic.decoder_spec("pamld", [words(7, 4), words(9, 4)], ["0:23:27", "1:22:26"]),
                synthetic.decoder_spec("mdd", [words(7, 4), words(9, 4)], ["0:27:31", "1:26:30"]),
                synthetic.decoder_spec("mdd", [words(2, 1)], ["0:6:7"]),
                synthetic.decoder_spec("pamld", [words(12, 12)], ["0:31:43"]),
                synthetic.decoder_spec("mdd", [words(12, 20)], ["1:30:50"])]
    molecular = [{"algorithm": "naive", "transform": {"token": ["0::6"]}},
                 synthetic.decoder_spec("pamld", [words(16, 5)], ["1::5"]),
                 synthetic.decoder_spec("mdd", [words(16, 5)], ["0:5:10"]),
                 {"algorithm": "naive", "transform": {"token": ["1:5:9"]}},
                 synthetic.decoder_spec("pamld", [words(20, 18)], ["0:40:58"]),
                 synthetic.decoder_spec("mdd", [words(3, 9)], ["1:50:59"])]
    spec = {"input segment cardinality": 2,
            "sample": synthetic.decoder_spec("pamld", [words(24, 8)], ["0:7:15"]),
            "cellular": cellular, "molecular": molecular}
    job = jobspec.compile_job(spec)
    assert 1 + len(job["cellular"]) + len(job["molecular"]) == 16
    n = 5000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 8)
    flag = (rng.random(n) < 0.2).astype(np.uint8)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, pieces=2)


@pytest.mark.parametrize("length", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 32])
def test_reads_without_information_sit_on_the_noise_threshold(oracle, torch_cuda, length):
    """A read of N bases (or of bases outside ACGT) costs the uniform quality at every position against every barcode:
    P(r|b) = pow(base, L * 10 log10 4), mathematically 4^-L = the default random barcode probability, and the noise
    filter `P > rbp` (pamld.cpp:96) is decided by the last bit of pow().  Index reads of all N are common in real
    runs, so the in-kernel power must round these arguments the way libm does — for every length, and for partial
    N runs and quality 0 positions around them."""
    rng = np.random.default_rng(length)
    count = min(12, 4 ** length)
    words = synthetic.barcode_set(rng, count, length, min_distance=1)
    job = jobspec.compile_job({"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words], ["0::%d" % length], noise=0.02)})
    n = 4096
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 5)
    code[: n // 4] = 15                                                     # all N
    quality[: n // 4] = rng.choice([2, 12, 25, 37, 40], size=(n // 4, 1)).astype(np.uint8)
    code[n // 4: n // 2] = rng.choice([15, 3, 5, 6, 7, 9, 14], size=(n // 4, length)).astype(np.uint8)      # nothing in ACGT
    some = rng.random((n // 4, length)) < 0.5                                # N runs among real bases
    block = code[n // 2: 3 * n // 4]
    block[some] = 15
    quality[n // 8: n // 4][rng.random((n // 8, length)) < 0.3] = 0          # quality 0 positions cost nothing
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)


@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
@pytest.mark.parametrize("name", ["U96", "U384", "V384", "S384", "S192", "L12", "L16"])
def test_unique_dual_index_pairs(oracle, torch_cuda, algorithm, name):
    """96 / 384 PAIRS of 8 + 8 nt in one two segment decoder (not every combination): the 16 nt product form / fused
    kernels, 384 pairs past the workgroup histogram (counters rebuilt from the records); 384 pairs of 10 + 10 nt (the
    two word product form kernel) and 384 single 8 nt indexes."""
    job, recipe = synthetic.config_job(name, algorithm)
    n = 20000 if name != "V384" else 12000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets, pieces=2)


@pytest.mark.parametrize("length,counts", [(8, (24, 20)), (9, (20, 24)), (10, (24, 20)), (8, (96, 96)), (10, (128, 17)),
                                           # round 6: a segment of exactly as many words as the lists hold (every word listed, nothing left to
                                           # bound), and sets of fewer than 512 combinations, which the product form over all of them took
                                           (8, (24, 16)), (8, (16, 16)), (8, (17, 17)), (9, (8, 40)), (10, (40, 8)), (10, (8, 8))])
def test_combinatorial_sets_in_the_streaming_kernel(oracle, torch_cuda, length, counts):
    """Dual index sets of 8 to 10 nt words holding every combination, past the workgroup histogram (340 bins): lists and
    factorised scan in one persistent kernel (pamld_cross_stream.hpp) — per segment the lookup, the listed words, the
    bound; three rings by what the second pass has to scan.  N calls, qualities of 0 and of 128 and more, uniform
    qualities (exact ties), short observations; first in the chain with and without a flag plane and behind another
    decoder (the state word); a batch in two pieces; both modes."""
    rng = np.random.default_rng(4242 + 100 * length + counts[0])
    segments = [synthetic.barcode_set(rng, c, length, min_distance=3) for c in counts]
    front = synthetic.barcode_set(rng, 5, 8, min_distance=3)
    spec = {"input segment cardinality": 3,
            "cellular": [synthetic.decoder_spec("pamld", [front], ["2::8"]),
                         synthetic.decoder_spec("pamld", segments, ["0::%d" % length, "1::%d" % length], **{"confidence threshold": 0.9})],
            "sample": synthetic.decoder_spec("pamld", segments, ["0::%d" % length, "1::%d" % length])}
    job = capi.compile_job(spec)
    n = 1 << 13
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 70 + length)
    noisy = rng.random(code.shape) < 0.01
    code[noisy] = np.array([15, 0, 3], dtype=np.uint8)[rng.integers(0, 3, size=int(noisy.sum()))]
    quality[rng.random(quality.shape) < 0.01] = 0
    quality[rng.random(quality.shape) < 0.002] = 200
    quality[: n // 8] = 30                                   # uniform qualities: exact ties between equidistant words
    for i in range(2 * length):                              # short observations: zero filled tail
        code[n // 2 + i, i:2 * length] = 0
        quality[n // 2 + i, i:2 * length] = 0
    flag[:] = rng.random(n) < 0.1
    s, engine, got, words = assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
    assert int((got[0]["status"] == 0).sum()) > n // 2       # most reads decode
    assert_parity(oracle, torch_cuda, job, code, quality, None, offsets, mode=capi.MODE_FAST, pieces=2)


@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
@pytest.mark.parametrize("counts", [(128, 3), (129, 3), (2, 128)])
def test_combinatorial_size_extremes(oracle, torch_cuda, algorithm, counts):
    """128 distinct words per segment is what the factorised kernels keep; one more and the codec takes the general
    kernels (384 / 387 / 256 barcodes, all past the workgroup histogram)."""
    rng = np.random.default_rng(sum(counts))
    segments = [synthetic.barcode_set(rng, counts[0], 8, min_distance=1), synthetic.barcode_set(rng, counts[1], 8, min_distance=1)]
    job = jobspec.compile_job({"input segment cardinality": 2, "sample": synthetic.decoder_spec(algorithm, segments, ["0::8", "1::8"])})
    n = 4096
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 6)
    assert_parity(oracle, torch_cuda, job, code, quality, flag, offsets)
