"""CPU-side checks of the C ABI library: it loads, exports every declared symbol, parses the compiled job,
extracts observations like Rule::apply, and writes the report like the reference.  No compute call here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import bdggg
from pheniqs_amd import DecoderSet, PheniqsError, capi
import jobspec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pheniqs_amd.h")).read()
    declared = set(re.findall(r"\b(phq_[a-z_]+)\s*\(", header))
    assert declared == set(capi.SYMBOLS)
    library = capi.load_library()
    for name in declared:
        assert hasattr(library, name), name


def test_create_reports_configuration_errors_with_reference_codes():
    job = jobspec.compile_job(bdggg.user_job())
    job["sample"]["algorithm"] = "bogus"
    with pytest.raises(PheniqsError) as e:
        DecoderSet(job, device=capi.DEVICE_NONE)
    assert e.value.code == 3 and "unsupported sample decoder algorithm bogus" in str(e.value)
    with pytest.raises(PheniqsError) as e:
        DecoderSet("{ not json", device=capi.DEVICE_NONE)
    assert e.value.code == 3
    job = jobspec.compile_job(bdggg.user_job())
    del job["sample"]["undetermined"]
    with pytest.raises(PheniqsError) as e:
        DecoderSet(job, device=capi.DEVICE_NONE)
    assert "classifier must declare an undetermined element" in str(e.value)      # classifier.cpp:96
    job = jobspec.compile_job(bdggg.user_job())
    job["sample"]["transform"]["token"] = ["1:8"]
    with pytest.raises(PheniqsError) as e:
        DecoderSet(job, device=capi.DEVICE_NONE)
    assert "illegal token syntax" in str(e.value)                                   # transform.cpp:166
    # compute without a device must fail loudly, never fall back
    host_only = DecoderSet(jobspec.compile_job(bdggg.user_job()), device=capi.DEVICE_NONE)
    with pytest.raises(PheniqsError):
        host_only.stats_export()


def test_layout_of_reference_compiled_job():
    """The library consumes the reference's own compiled ontology (valid/compile_annotated.out) unchanged."""
    s = DecoderSet(json.dumps(bdggg.compiled_reference_job()), device=capi.DEVICE_NONE)
    assert (s.n_decoder, s.n_result, s.nucleotide_width, s.plane_width) == (3, 2, 24, 24)
    kinds = [(d["kind"], d["algorithm"], d["n_barcode"], d["result_slot"], d["multiplexing"]) for d in s.decoders]
    assert kinds == [(0, 1, 5, 0, 1), (1, 3, 0, -1, 0), (2, 1, 5, 1, 0)]
    assert s.stats_words == 2 + 12 * (6 + 1 + 6)


def _segments(reads):
    out = []
    for k in range(3):
        codes = [r["segments"][k]["code"] for r in reads]
        quals = [r["segments"][k]["quality"] for r in reads]
        length = np.array([len(c) for c in codes], dtype=np.int32)
        offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int64)
        out.append({"code": np.concatenate(codes), "quality": np.concatenate(quals), "offset": offset, "length": length})
    return out


def test_stage_reads_matches_rule_apply_oracle(oracle):
    job = jobspec.compile_job(bdggg.user_job())
    reads = bdggg.load_reads(oracle.ascii_to_bam)
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    flags = np.array([1 if r["qcfail"] else 0 for r in reads], dtype=np.uint8)
    staged = s.stage_reads(_segments(reads), flags)
    assert staged["incoming"] == (250, 243)                     # golden "incoming" block
    assert staged["n"] == 248                                   # two reads fail min input length [0, 0, 22]
    kept = [reads[i] for i in staged["kept"]]
    engine = oracle.OracleJob(job)
    rows_code, rows_quality = engine.stage([[(x["code"], x["quality"]) for x in r["segments"]] for r in kept])
    offsets = [engine.decoder_offset(i) for i in range(3)]
    n = staged["capacity"]
    for d, offset in zip(s.decoders, offsets):
        L, stride, prefix = d["nucleotide_cardinality"], d["plane_stride"], d["plane_prefix"]
        plane_code = staged["base"][n * prefix: n * (prefix + stride)].reshape(n, stride)[:248]
        plane_quality = staged["quality"][n * prefix: n * (prefix + stride)].reshape(n, stride)[:248]
        assert np.array_equal(plane_code[:, :L], rows_code[:, offset:offset + L])
        assert np.array_equal(plane_quality[:, :L], rows_quality[:, offset:offset + L])
    assert np.array_equal(staged["flag"][:248], flags[staged["kept"]])


def test_stage_reads_reverse_complement_negative_tokens_and_short_reads(oracle):
    spec = {"input segment cardinality": 2,
            "cellular": [{"algorithm": "pamld", "transform": {"token": ["0:-6:", "1:2:5", "1:-3:"], "knit": ["~0:1", "~2"]},
                          "codec": {"@A": {"barcode": ["ACGTACGTA", "TTT"]}, "@B": {"barcode": ["CCCCAAAAC", "GGG"]}}}]}
    job = jobspec.compile_job(spec)
    rng = np.random.default_rng(7)
    reads = []
    for length0, length1 in [(20, 12), (4, 12), (20, 3), (0, 0), (6, 5), (3, 2)]:
        segs = []
        for length in (length0, length1):
            segs.append((BAM[rng.integers(0, 5, size=length)], rng.integers(0, 42, size=length).astype(np.uint8)))
        reads.append(segs)
    engine = oracle.OracleJob(job)
    rows_code, rows_quality = engine.stage(reads)
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    segments = []
    for k in range(2):
        length = np.array([len(r[k][0]) for r in reads], dtype=np.int32)
        offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int64)
        segments.append({"code": np.concatenate([r[k][0] for r in reads]), "quality": np.concatenate([r[k][1] for r in reads]),
                         "offset": offset, "length": length})
    staged = s.stage_reads(segments, None)
    n, d = staged["capacity"], s.decoders[0]
    plane = staged["base"][: n * d["plane_stride"]].reshape(n, d["plane_stride"])
    assert np.array_equal(plane[:, :12], rows_code)
    assert not plane[:, 12:].any()
    plane_q = staged["quality"][: n * d["plane_stride"]].reshape(n, d["plane_stride"])
    assert np.array_equal(plane_q[:, :12], rows_quality)


BAM = np.array([1, 2, 4, 8, 15], dtype=np.uint8)


def _words_from_oracle(s, engine):
    """Encode the oracle's counters in the library's integer counter layout (include/pheniqs_amd.h)."""
    words = np.zeros(s.stats_words, dtype=np.uint64)
    count, pf_count = engine.outgoing()
    words[0], words[1] = count - pf_count, pf_count
    for k, d in enumerate(s.decoders):
        option = engine.options(k)
        for b in range(len(option)):
            o = option[b]
            base = d["stats_offset"] + b * 12
            fail, passed = base, base + 6
            words[passed + 0] = o["pf_count"]; words[fail + 0] = o["count"] - o["pf_count"]
            words[passed + 1] = o["accumulated_pf_distance"]; words[fail + 1] = o["accumulated_distance"] - o["accumulated_pf_distance"]
            total = int(round(float(o["accumulated_confidence"]) * 2 ** 52))
            passed_fixed = int(round(float(o["accumulated_pf_confidence"]) * 2 ** 52))
            failed_fixed = total - passed_fixed
            words[passed + 2], words[passed + 3] = passed_fixed & (2 ** 26 - 1), passed_fixed >> 26
            words[fail + 2], words[fail + 3] = failed_fixed & (2 ** 26 - 1), failed_fixed >> 26
            words[fail + 4] = o["low_conditional_confidence_count"]
            words[fail + 5] = o["low_confidence_count"]
    return words


def test_report_writer_reproduces_golden_report(oracle):
    """finalize + encode + print_json of the library on the reference run's counters == valid/annotated.err."""
    job = jobspec.compile_job(bdggg.user_job())
    reads = bdggg.load_reads(oracle.ascii_to_bam)
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    flags = np.array([1 if r["qcfail"] else 0 for r in reads], dtype=np.uint8)
    staged = s.stage_reads(_segments(reads), flags)
    kept = [reads[i] for i in staged["kept"]]
    engine = oracle.OracleJob(job)
    rows_code, rows_quality = engine.stage([[(x["code"], x["quality"]) for x in r["segments"]] for r in kept])
    engine.classify(rows_code, rows_quality, flags[staged["kept"]])
    words = _words_from_oracle(s, engine)
    for k in range(3):
        mine, theirs = s.stats_options(words, k), engine.options(k)
        for name in mine.dtype.names:
            if name.startswith("accumulated_") and name.endswith("confidence"):
                assert np.allclose(mine[name], theirs[name], rtol=0, atol=1e-13)
            else:
                assert np.array_equal(mine[name], theirs[name]), name
    text = s.report(words, staged["incoming"])
    report = json.loads(text)
    golden = bdggg.golden_report()

    def compare(mine, theirs, path):
        if isinstance(mine, dict):
            for key, value in mine.items():
                assert key in theirs, path + "/" + key
                compare(value, theirs[key], path + "/" + key)
        elif isinstance(mine, list):
            assert len(mine) == len(theirs), path
            for i, (m, t) in enumerate(zip(mine, theirs)):
                compare(m, t, "%s[%d]" % (path, i))
        elif isinstance(theirs, float):
            assert abs(mine - theirs) <= 1e-15 + 1e-18, (path, mine, theirs)
        else:
            assert mine == theirs, (path, mine, theirs)

    compare(report, golden, "")
    assert set(report.keys()) == set(golden.keys())

    # the prior estimator (phq_adjust_prior, classifier.h:125-160) on the user level document (its repository and
    # `base` references are resolved first, like Transcode::assemble) writes the report's estimated priors back,
    # at 15 truncated decimals like the report
    raw = bdggg.raw_job()
    adjusted = s.adjust_prior(words, raw)
    assert set(adjusted.keys()) == set(raw.keys()) - {"decoder"} and "base" not in adjusted["sample"]
    for mine_node, theirs_node in ((adjusted["sample"], golden["sample"]), (adjusted["cellular"][0], golden["cellular"][0])):
        assert mine_node["noise"] == theirs_node["estimated noise"]
        by_barcode = {tuple(r["barcode"]): r for r in theirs_node["classified"]}
        assert set(mine_node["codec"]) == set(raw["sample"]["codec"])
        for key, record in mine_node["codec"].items():
            barcode = tuple(raw["decoder"]["BDGGG_sample"]["codec"][key]["barcode"])
            assert record["concentration"] == by_barcode[barcode]["estimated concentration"], key
            assert record["LB"] == raw["sample"]["codec"][key]["LB"]
    assert adjusted["molecular"][0]["noise"] == 0
    # layout: sorted keys, 4 space indent, integers bare — same lines as the golden file for a count only block
    golden_text = open(os.path.join(bdggg.HERE, "annotated.report.json")).read()
    for block in ("incoming", "outgoing"):
        pattern = r'    "%s": \{.*?\n    \}' % block
        assert re.search(pattern, text, re.S).group(0) == re.search(pattern, golden_text, re.S).group(0)
    mine_lines = [l for l in text.splitlines() if '"estimated' not in l and "average" not in l]
    assert mine_lines[0] == "{" and mine_lines[-1] == "}"


def _io_job(**changes):
    job = bdggg.raw_job()
    job["base input url"] = "/data/in"
    job["base output url"] = "/data/out"
    job.update(changes)
    return job


def test_transcode_plan_of_the_reference_job():
    """phq_transcode_plan on the reference's own compiled job: the feeds of its `feed` section (valid/compile_annotated.out:
    three FASTQ inputs of resolution 1, one SAM output of resolution 2) and every channel routed to that output."""
    reference = bdggg.compiled_reference_job()
    plan = DecoderSet(json.dumps(reference), device=capi.DEVICE_NONE).transcode_plan()
    theirs = reference["feed"]
    assert [(f["url"], f["resolution"]) for f in plan["input feed"]] == [(f["url"].split("?")[0], f["resolution"]) for f in theirs["input feed"]]
    assert plan["input feed by segment"] == [[0, 0], [1, 0], [2, 0]]
    assert [(f["url"], f["format"], f["compression"], f["resolution"]) for f in plan["output feed"]] == \
           [(f["url"].split("?")[0], "sam", "none", f["resolution"]) for f in theirs["output feed"]] == [("/dev/stdout", "sam", "none", 2)]
    assert len(plan["channel"]) == 6 and all(c["output feed by segment"] == [0, 0] and not c["filter outgoing qc fail"] for c in plan["channel"])


def test_transcode_plan_feeds_channels_and_errors():
    """Interleaved input (a URL stated k times = one feed of resolution k, transcode.cpp:693-701), outputs per channel with
    their formats and compression by extension or query (url.cpp:280-417), the consistency checks of compile_output."""
    job = _io_job(input=["lane1.fastq.gz", "lane1.fastq.gz", "index.fq"], output=["rest.sam"])
    codec = job["sample"]["codec"]
    codec["@AGGCAGAA"]["output"] = ["a_1.fastq.gz", "a_2.fastq.gz"]
    codec["@CGTACTAG"]["output"] = ["b.fq?compression=gz&level=7"]
    codec["@GGACTCCT"]["output"] = ["/abs/c.txt?format=fastq"]
    codec["@GGACTCCT"]["filter outgoing qc fail"] = True
    plan = DecoderSet(json.dumps(job), device=capi.DEVICE_NONE).transcode_plan()
    assert plan["input feed"] == [{"index": 0, "url": "/data/in/lane1.fastq.gz", "resolution": 2}, {"index": 1, "url": "/data/in/index.fq", "resolution": 1}]
    assert plan["input feed by segment"] == [[0, 0], [0, 1], [1, 0]]
    feeds = {f["url"]: f for f in plan["output feed"]}
    assert (feeds["/data/out/rest.sam"]["format"], feeds["/data/out/rest.sam"]["compression"], feeds["/data/out/rest.sam"]["resolution"]) == ("sam", "none", 2)
    assert (feeds["/data/out/a_1.fastq.gz"]["format"], feeds["/data/out/a_1.fastq.gz"]["compression"], feeds["/data/out/a_1.fastq.gz"]["resolution"]) == ("fastq", "gz", 1)
    assert (feeds["/data/out/b.fq"]["format"], feeds["/data/out/b.fq"]["compression"], feeds["/data/out/b.fq"]["compression level"], feeds["/data/out/b.fq"]["resolution"]) == ("fastq", "gz", 7, 2)
    assert (feeds["/abs/c.txt"]["format"], feeds["/abs/c.txt"]["resolution"]) == ("fastq", 2)
    channel = plan["channel"]                                  # undetermined, then the codec in key order
    rest = feeds["/data/out/rest.sam"]["index"]
    assert channel[0]["output feed by segment"] == [rest, rest]
    assert channel[1]["output feed by segment"] == [feeds["/data/out/a_1.fastq.gz"]["index"], feeds["/data/out/a_2.fastq.gz"]["index"]]
    assert channel[2]["output feed by segment"] == [feeds["/data/out/b.fq"]["index"]] * 2
    assert channel[3]["filter outgoing qc fail"] and not channel[4]["filter outgoing qc fail"]
    assert channel[4]["output feed by segment"] == [rest, rest] == channel[5]["output feed by segment"]
    for changes, text in (({"output": ["a.sam", "b.sam", "c.sam"]}, "incorrect number of output URLs in channel"),
                          ({"output": ["/dev/stdin"]}, "output stream can not be set to standard input"),
                          ({"output": ["out.cram"]}, "htslib"),
                          ({"output": ["out.fastq.xz"]}, "not supported"),
                          ({"input": ["a.bam", "b.fastq", "c.fastq"]}, "only FASTQ input")):
        with pytest.raises(PheniqsError) as caught:
            DecoderSet(json.dumps(_io_job(**changes)), device=capi.DEVICE_NONE).transcode_plan()
        assert text in str(caught.value), (changes, str(caught.value))
    job = _io_job(output=["rest.sam"])
    job["sample"]["codec"]["@AGGCAGAA"]["output"] = ["x.fastq", "x.fastq"]
    job["sample"]["codec"]["@CGTACTAG"]["output"] = ["x.fastq", "y.fastq"]
    with pytest.raises(PheniqsError) as caught:
        DecoderSet(json.dumps(job), device=capi.DEVICE_NONE).transcode_plan()
    assert "inconsistent resolution" in str(caught.value)


@pytest.mark.parametrize("value,expected", [
    (0.0, "0.0"), (1.0, "1.0"), (0.972, "0.972"), (0.18 * (1 - 0.015), "0.177299999999999"),
    (1.0 / 4 ** 8, "0.000015258789062"), (0.951612903225806451, "0.951612903225806"), (248.0, "248.0"),
    (0.5, "0.5"), (1e-20, "0.0"), (0.049180327868852458, "0.049180327868852"), (123456.789, "123456.789"),
    (2.0 / 3.0, "0.666666666666666"), (0.1 + 0.2, "0.3"), (1e21, "1e21"), (1.5e22, "1.5e22"), (-0.25, "-0.25"),
])
def test_format_double_is_rapidjson_truncation(value, expected):
    """Values taken from the golden report / compiled job, plus the branches of rapidjson's Prettify."""
    assert capi.format_double(value, 15) == expected


def test_host_side_under_sanitizers(tmp_path):
    """The library's CPU side (JSON reader, job model, PhredScale, report writer) under ASan + UBSan; GPU
    sanitizers are not available on the target pool."""
    import json
    import os
    import subprocess

    from pheniqs_amd import synthetic

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        # no sanitizer runtime on the machine is a reason to skip; an error in the library's own sources (a stub of
        # check/no_device.cpp that no longer matches, a missing include) is not
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    paths = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "BDGGG", "compile_annotated.json")]
    for name in ("C2", "C5"):
        job = synthetic.config_job(name, "mdd")[0]
        path = tmp_path / (name + ".json")
        path.write_text(json.dumps(job))
        paths.append(str(path))
    raw = tmp_path / "raw.json"                          # uncompiled: decoder repository and base references
    raw.write_text(json.dumps(bdggg.raw_job()))
    paths.append(str(raw))
    ran = subprocess.run([os.path.join(csrc, "build", "host_check_asan")] + paths, capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout + ran.stderr
    assert "host check ok" in ran.stdout


def test_transcoder_reader_and_writers_under_sanitizers(tmp_path):
    """csrc/transcoder.cpp — mapped input, 32 byte line scans, raw pointer record writers — under ASan + UBSan, driven with
    damaged FASTQ files (fastq.h:427-440: truncated records, broken separators, names out of sync) and with result planes
    the check injects (auxiliary.cpp:327-354 is the writers' side): an error code or a result every time, nothing for the
    sanitizers to report.  The test's own exact match planes stand where the GPU's would be; the product path is not touched."""
    import json
    import subprocess

    import numpy as np

    import bench
    from pheniqs_amd import synthetic

    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        # no sanitizer runtime on the machine is a reason to skip; an error in the library's own sources (a stub of
        # check/no_device.cpp that no longer matches, a missing include) is not
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    check = os.path.join(csrc, "build", "transcoder_check_asan")
    # the reference's own three segment test files, 250 records each: 260 damaged runs
    work = tmp_path / "bdggg"
    work.mkdir()
    job = work / "job.json"
    job.write_text(json.dumps(bdggg.user_job()))
    inputs = [os.path.join(here, "golden", "BDGGG", "BDGGG_s0%d.fastq" % k) for k in (1, 2, 3)]
    ran = subprocess.run([check, str(job), "260", str(work)] + inputs, capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout + ran.stderr
    assert "261 runs" in ran.stdout
    # a two segment job over 3 000 synthetic records (batches of 37 records: many batches, the read ahead of the readers),
    # every barcode a channel with FASTQ outputs of its own for the rounds that take the job's directives: 140 runs
    work = tmp_path / "c2"
    work.mkdir()
    spec = synthetic.config_spec("C2", "mdd")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    spec["enable quality control"] = True                # the accumulators' hook in the record threads (qc.cpp) runs under the sanitizers too
    spec["base output url"] = str(work)
    for number, record in enumerate(spec["sample"]["codec"].values()):
        if number < 5:
            record["output"] = ["channel_%d_s1.fastq" % number, "channel_%d_s2.fastq.gz" % number]
        elif number < 8:
            record["output"] = ["channel_%d.bam" % number]          # unaligned BAM: bam_record.hpp and the BGZF writer under the sanitizers
    spec["sample"]["output"] = ["rest.sam"]
    compiled = capi.compile_job(spec)
    code, quality, flag, _ = synthetic.generate_rows(compiled, 3000, 11)
    letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    rng = np.random.default_rng(5)
    template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(3000, 30), dtype=np.uint8)]
    paths = []
    for k, (text, qual) in enumerate(((template, np.full((3000, 30), ord("F"), dtype=np.uint8)), (letters[code], (np.minimum(quality, 93) + 33).astype(np.uint8)))):
        path = str(work / ("segment_in_%d.fastq" % k))
        bench.write_fastq(path, k + 1, text, qual, flag)
        paths.append(path)
    spec["input"] = paths                              # (the rounds that take the job's directives need a job that states its input: the check puts its damaged copies in their place)
    job = work / "job.json"
    job.write_text(json.dumps(spec))
    ran = subprocess.run([check, str(job), "140", str(work)] + paths, capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout + ran.stderr
    results = int(ran.stdout.split(" runs, ")[1].split(" results")[0])
    assert "141 runs" in ran.stdout and results >= 20, ran.stdout
    # the channels' own files were written: FASTQ, gzip members, and BAM files that read back block by block
    import bam_reader
    assert os.path.getsize(work / "channel_0_s1.fastq") > 0 and open(work / "channel_0_s2.fastq.gz", "rb").read(2) == b"\x1f\x8b"
    for number in (5, 6, 7):
        _, records, blocks = bam_reader.records(str(work / ("channel_%d.bam" % number)))
        assert blocks >= 2


def test_a_damaged_quality_control_image_is_refused_before_anything_is_allocated():
    """phq_qc_merge on an image gathered from another rank: a header that promises more cells or cycles than the image
    holds, or an average that is not a number, is an error — never an allocation sized by the header alone."""
    s = DecoderSet(jobspec.compile_job(bdggg.user_job()), device=capi.DEVICE_NONE)
    cell_fixed = 2 + 16 + 3 + 42                                     # capacity, shortest, 16 nucleic counts, 3 average fields, 42 distribution words (qc.hpp)
    promised = np.array([1 << 20, 64] + [0] * 100, dtype=np.uint64)              # 67 M cells in a hundred words
    huge_cell = np.array([1, 1, 1 << 20, 0] + [0] * (cell_fixed + 10), dtype=np.uint64)      # one cell of 2^20 cycles
    for image in (promised, huge_cell, np.array([3], dtype=np.uint64)):
        with pytest.raises(PheniqsError):
            s.qc_merge(image)
    s.close()


def test_pack_planes_is_lossless_on_the_host():
    """phq_pack_planes (no device needed): ordinal << 6 | quality for A, C, G, T up to quality 62, 0x7F for (0, 0), 0x3F and a list
    entry (read << 32 | column << 16 | code << 8 | quality, in read order) for everything else; unpacked here in numpy."""
    from pheniqs_amd import synthetic
    spec = synthetic.config_spec("C4", "pamld")
    job = capi.compile_job(spec)
    n = 5000
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 17)
    rng = np.random.default_rng(2)
    quality[rng.random(quality.shape) < 0.01] = 70
    code[rng.random(code.shape) < 0.01] = 5
    s = DecoderSet(spec, device=capi.DEVICE_NONE)
    base, qual = s.planes_from_rows(code, offsets), s.planes_from_rows(quality, offsets)
    planes, escapes = s.pack_planes(n, base, qual)
    codes = (1 << (planes >> 6)).astype(np.uint8)
    values = (planes & 63).astype(np.uint8)
    nothing = planes == 0x7F
    codes[nothing] = 0
    values[nothing] = 0
    listed = planes == 0x3F
    assert int(listed.sum()) == escapes.size > 0
    reads = (escapes >> np.uint64(32)).astype(np.int64)
    assert np.all(np.diff(reads) >= 0)
    column = ((escapes >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
    for d in s.decoders:
        inside = (column >= d["plane_prefix"]) & (column < d["plane_prefix"] + d["plane_stride"])
        at = n * d["plane_prefix"] + reads[inside] * d["plane_stride"] + column[inside] - d["plane_prefix"]
        assert np.all(listed[at])
        codes[at] = ((escapes[inside] >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
        values[at] = (escapes[inside] & np.uint64(0xFF)).astype(np.uint8)
    assert np.array_equal(codes, base) and np.array_equal(values, qual)
    # a list too small: only the count comes back
    needed = ctypes.c_int64(0)
    packed = np.zeros_like(base)
    few = np.zeros(4, dtype=np.uint64)
    assert s.lib.phq_pack_planes(s.handle, n, n, base.ctypes.data, qual.ctypes.data, packed.ctypes.data, few.ctypes.data, 4, ctypes.byref(needed)) == 0
    assert needed.value == escapes.size and not few.any()
    s.close()


def _bgzf(text, block=65280):
    """bgzip's container (SAM specification 4.1): blocks of at most 64 KiB, each a gzip member whose extra subfield 'B' 'C'
    holds its size, an empty block at the end — written here from the specification, not by htslib."""
    import struct
    import zlib
    out = bytearray()
    pieces = [text[k:k + block] for k in range(0, len(text), block)] + [b""]
    for piece in pieces:
        compressor = zlib.compressobj(6, zlib.DEFLATED, -15)
        data = compressor.compress(piece) + compressor.flush()
        size = 12 + 6 + len(data) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
        out += data + struct.pack("<II", zlib.crc32(piece), len(piece))
    return bytes(out)


def test_gzip_members_inflate_in_parallel_to_the_same_text(tmp_path):
    """gzip_members.hpp under ASan + UBSan: BGZF blocks and plain concatenated members come back as the text they hold, read in
    pieces of uneven sizes across batches; a single member stream is left to gzread; a member cut short or with a damaged trailer
    is an IO error, never wrong text (fastq.h:427-440 reads such files through htslib's bgzf)."""
    import gzip
    import struct
    import subprocess
    import zlib

    import numpy as np

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    check = os.path.join(csrc, "build", "gzip_members_check_asan")
    rng = np.random.default_rng(8)
    records = []
    for k in range(60000):
        length = int(rng.integers(30, 90))
        records.append(b"@M:1:FC:1:1101:%07d:%05d 1:N:0:1\n" % (k, k % 77777) + bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), length, p=[.24, .25, .25, .25, .01])) + b"\n+\n" +
                       bytes(rng.choice(np.frombuffer(b"#-:F", dtype=np.uint8), length)) + b"\n")
    text = b"".join(records)
    expected = tmp_path / "text.fastq"
    expected.write_bytes(text)

    def run(data, compare=True):
        path = tmp_path / "input.gz"
        path.write_bytes(data)
        return subprocess.run([check, str(path)] + ([str(expected)] if compare else []), capture_output=True, text=True)

    blocked = run(_bgzf(text))
    assert blocked.returncode == 0 and " blocked 1 " in blocked.stdout, blocked.stdout + blocked.stderr
    assert int(blocked.stdout.split()[1]) == -(-len(text) // 65280)
    assert blocked.stdout.split()[-3] == str(-(-len(text) // 65280)) and blocked.stdout.split()[-1] == "0"      # every block through the one shot decoder (fast_inflate.hpp), none through zlib
    cuts = [0, 1000, 1001, len(text) // 3, len(text) // 2, len(text) - 5, len(text)]        # members of very different sizes
    members = b"".join(gzip.compress(text[a:b], level) for (a, b), level in zip(zip(cuts[:-1], cuts[1:]), (1, 9, 6, 1, 5, 9)))
    plain = run(members)
    assert plain.returncode == 0 and "members 6 blocked 0 " in plain.stdout, plain.stdout + plain.stderr
    assert run(gzip.compress(text, 1)).returncode == 3                                      # one member: the caller's gzread path
    assert run(text).returncode == 3                                                        # not gzip at all
    damaged = bytearray(members)
    damaged[-2] ^= 0x55                                                                     # the size the last member states
    assert run(bytes(damaged)).returncode == 4
    damaged = bytearray(members)
    damaged[len(gzip.compress(text[:1000], 1)) + 40] ^= 0xff                                # inside the second member's deflate data
    assert run(bytes(damaged)).returncode == 4
    assert run(members[:-9]).returncode == 4                                                # cut short
    cut = _bgzf(text)
    assert run(cut[:len(cut) // 2]).returncode in (3, 4)                                    # half a BGZF file: not taken as blocks, or an error
    # two lanes' files joined with cat: members past the reader's limit are left to the caller's streaming path ...
    large = gzip.compress(b"A" * ((64 << 20) + 1), 1) + gzip.compress(text[:1000], 6)
    assert run(large, compare=False).returncode == 3
    # ... and a member of 4 GiB and more of text, whose trailer holds its size modulo 2^32, is an error at its first read, never
    # other text (the transcoder then reads the file through gzread from its start)
    packer = zlib.compressobj(1, zlib.DEFLATED, 31)
    piece = bytes(64 << 20)
    wrapped = b"".join(packer.compress(piece) for _ in range(64)) + packer.compress(b"@" * 1000) + packer.flush()
    assert struct.unpack("<I", wrapped[-4:])[0] == 1000 and len(wrapped) < (64 << 20)
    assert run(wrapped + gzip.compress(text[:1000], 6), compare=False).returncode == 4


def test_a_single_member_gzip_stream_inflates_to_the_same_text(tmp_path):
    """gzip_stream.hpp under ASan + UBSan: fast_inflate.hpp's decoder in streaming form — 1 MiB of work buffer behind 32 KiB of
    history, resumed inside blocks — gives the text of single member files of every zlib level and strategy (stored, fixed and
    dynamic blocks; FASTQ text, bytes that do not compress, long runs), of a few large members one behind another, and of a file
    with bytes behind its last member (gzread ends the text there too); a block handed to zlib part way (the decoder's refusal,
    forced at block 0, 1, 7) continues from the history and the bit offset of that block; damaged, cut and CRC-less files are IO
    errors (fastq.h:427-440 reads these files through htslib)."""
    import gzip
    import subprocess
    import zlib

    import numpy as np

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    check = os.path.join(csrc, "build", "gzip_stream_check_asan")
    rng = np.random.default_rng(12)
    records = []
    for k in range(24000):                      # 3.4 MB: several work buffers of a MiB
        length = int(rng.integers(30, 90))
        records.append(b"@M:1:FC:1:1101:%07d:%05d 1:N:0:1\n" % (k, k % 77777) + bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), length, p=[.24, .25, .25, .25, .01])) + b"\n+\n" +
                       bytes(rng.choice(np.frombuffer(b"#-:F", dtype=np.uint8), length, p=[.03, .07, .1, .8])) + b"\n")
    fastq = b"".join(records)
    texts = {"fastq": fastq, "random": rng.integers(0, 256, (1 << 20) + 77777, dtype=np.uint8).tobytes(), "run": b"F" * ((1 << 20) + 5000) + fastq[:100000] + b"G" * 70000,
             "empty": b"", "byte": b"x"}

    def run(data, text, *more):
        path, expected = tmp_path / "input.gz", tmp_path / "text"
        path.write_bytes(data)
        expected.write_bytes(text)
        return subprocess.run([check, str(path), str(expected)] + [str(m) for m in more], capture_output=True, text=True)

    for name, text in texts.items():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                                (6, zlib.Z_HUFFMAN_ONLY), (3, zlib.Z_RLE), (9, zlib.Z_FILTERED)):
            if True:
                packer = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
                ran = run(packer.compress(text) + packer.flush(), text)
                assert ran.returncode == 0 and " zlib 0" in ran.stdout, (name, level, strategy, ran.stdout, ran.stderr[-500:])
    packed = gzip.compress(fastq, 6)
    for block in (0, 1, 7):
        ran = run(packed, fastq, block)
        assert ran.returncode == 0 and " zlib 1" in ran.stdout, (block, ran.stdout, ran.stderr[-500:])
    thirds = [fastq[:len(fastq) // 3], fastq[len(fastq) // 3:len(fastq) // 2], fastq[len(fastq) // 2:]]
    assert run(b"".join(gzip.compress(part, level) for part, level in zip(thirds, (1, 9, 5))), fastq).returncode == 0      # cat a.gz b.gz c.gz
    assert run(packed + bytes(1000), fastq).returncode == 0                                  # padding behind the member
    damaged = bytearray(packed)
    damaged[len(packed) // 2] ^= 0x10
    assert run(bytes(damaged), fastq).returncode in (4, 5)                                    # an error — or, rarely, other text that then fails its CRC: 4
    assert run(bytes(damaged), fastq).returncode == 4
    damaged = bytearray(packed)
    damaged[-6] ^= 0xff                                                                       # the CRC
    assert run(bytes(damaged), fastq).returncode == 4
    assert run(packed[:len(packed) // 2], fastq).returncode == 4                              # cut short
    assert run(packed[:-3], fastq).returncode == 4
    assert run(fastq, fastq).returncode == 3                                                  # not gzip

    # the same stream on four threads (gzip_stream.hpp, marked_decode): chunks that start at block headers found by search, their
    # matches into the unknown 32 KiB in front of them kept as markers and resolved when the chunk before is — the text is the same,
    # chunks are taken, and everything above holds: a block handed to zlib, members one behind another, damage, a cut file,
    # streams of fixed or stored blocks (no header the search takes: the one thread decodes them through)
    large = b"".join(records[k % len(records)][:-1] + b"%d\n" % k for k in range(120000))       # 17 MB of text, 4 MB packed
    for level in (1, 6, 9):
        ran = run(gzip.compress(large, level), large, -1, 4)
        assert ran.returncode == 0 and " zlib 0 " in ran.stdout and int(ran.stdout.split()[-3]) >= 3, (level, ran.stdout, ran.stderr[-500:])
    packed = gzip.compress(large, 6)
    for block in (0, 3, 9):                            # (of the stream's own thread: its count starts again behind every chunk taken)
        ran = run(packed, large, block, 4)
        assert ran.returncode == 0 and " zlib 1 " in ran.stdout, (block, ran.stdout, ran.stderr[-500:])
    half = len(large) // 2
    assert run(gzip.compress(large[:half], 4) + gzip.compress(large[half:], 7) + bytes(100), large, -1, 4).returncode == 0
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY):
        packer = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
        assert run(packer.compress(large) + packer.flush(), large, -1, 4).returncode == 0
    noise = rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes()
    assert run(gzip.compress(noise, 6), noise, -1, 4).returncode == 0
    for where in (len(packed) // 5, len(packed) // 2, len(packed) - 3000):
        damaged = bytearray(packed)
        damaged[where] ^= 0x04
        assert run(bytes(damaged), large, -1, 4).returncode == 4
    assert run(packed[:len(packed) * 2 // 3], large, -1, 4).returncode == 4


def test_the_gzip_members_the_library_writes_come_back_through_zlib_and_its_own_decoder():
    """csrc/fast_deflate.hpp (RFC 1951 / 1952) under ASan + UBSan: the compressor behind the transcoder's gzip outputs up to the
    reference's default level 5 — one hash probe a position, a match only where it is cheaper than its bytes as literals, dynamic
    codes per block, stored blocks where coding does not pay — on nothing, 1 to 1 000 bases, FASTQ text and its prefixes around the
    window and block boundaries, noise, a long run, short periods, noisy text, far matches, zeros, every byte value: every member
    comes back byte for byte through zlib's inflate AND through fast_inflate.hpp, with the CRC and the size its trailer states;
    on FASTQ text it is within 10 % of zlib level 5's size (fastq.h:144-194 writes such files through htslib's bgzf)."""
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    ran = subprocess.run([os.path.join(csrc, "build", "fast_deflate_check_asan")], capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout[-2000:] + ran.stderr[-2000:]
    match = re.search(r"members (\d+) ratio ([0-9.]+) \(zlib level 5: ([0-9.]+)\)", ran.stdout)
    assert match and int(match.group(1)) >= 33 and float(match.group(2)) > 0.9 * float(match.group(3)), ran.stdout


def test_the_one_shot_inflate_against_zlib_under_sanitizers():
    """csrc/fast_inflate.hpp (RFC 1951 / 1952) under ASan + UBSan: 800 gzip members — empty, one byte, FASTQ text, random bytes, a
    long run, short periods, noisy text, far matches; zlib levels 0 to 9, every strategy (stored, fixed and dynamic blocks) — come
    back byte for byte with nothing written past the text; 3 000 damaged members are refused or come back as the text they still
    hold, never as other text; the CRC by carry-less multiplication equals zlib's at every length around its block sizes."""
    import subprocess

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pheniqs_amd", "csrc")
    built = subprocess.run(["make", "-s", "-j", "6", "-C", csrc, "asan"], capture_output=True, text=True)
    if built.returncode != 0:
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    ran = subprocess.run([os.path.join(csrc, "build", "fast_inflate_check_asan")], capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout[-2000:] + ran.stderr[-2000:]
    assert "members 800 refused 0" in ran.stdout


def test_bam_records_from_sam_lines_read_back_field_by_field(tmp_path):
    """csrc/bam_record.hpp + the BGZF writer of csrc/fast_deflate.hpp without a device and without the transcoder around them
    (check/bam_record_check.cpp under ASan + UBSan): 6 000 unaligned SAM lines — names up to 254 characters, sequences of every
    length from none ('*') to 300 with IUPAC letters in either case, string / int32 / float tags, the floats stated as their bits
    the way the device states them for a BAM sink — come back through tests/bam_reader.py (written from the specification alone)
    field by field; lines this writer does not make are errors, not records."""
    import struct
    import subprocess
    import bam_reader
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "pheniqs_amd", "csrc")
    check = str(tmp_path / "bam_record_check_asan")
    built = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-Wall", "-Wno-unused-result",
                            os.path.join(csrc, "check", "bam_record_check.cpp"), "-o", check, "-lz"], capture_output=True, text=True)
    if built.returncode != 0:
        assert "libasan" in built.stderr or "libubsan" in built.stderr or "fsanitize" in built.stderr, built.stderr[-2000:]
        pytest.skip("sanitizer build unavailable: " + built.stderr[-200:])
    rng = np.random.default_rng(23)
    letters = "=ACMGRSVTWYHKDBNacgtn"
    expected, lines = [], []
    for k in range(6000):
        name = "M:%d:" % k + "x" * int(rng.integers(0, 40)) if k % 500 else "n" * 254
        length = 0 if k % 97 == 0 else int(rng.integers(1, 301))
        sequence = "".join(letters[i] for i in rng.integers(0, len(letters), length))
        quality = bytes(rng.integers(0, 94, length, dtype=np.uint8))
        flag = int(rng.choice([77, 141, 589, 653, 4]))
        tags = []
        if k % 3:
            tags.append(("RG", "Z", "group:%d" % (k % 7)))
        if k % 2:
            tags += [("BC", "Z", "ACGTNNAC-TTGA"), ("QT", "Z", "FF,:#FFF FFFF")]
        if k % 5 == 0:
            tags += [("FI", "i", k % 4 + 1), ("TC", "i", 4), ("XX", "i", -12345 - k)]
        for tag in ("XB", "XM", "XC"):
            if rng.random() < 0.5:
                tags.append((tag, "f", struct.unpack("<f", struct.pack("<f", float(rng.random()) * 10.0 ** -int(rng.integers(0, 12))))[0]))
        fields = [name, str(flag), "*", "0", "0", "*", "*", "0", "0", sequence or "*", "".join(chr(q + 33) for q in quality) or "*"]
        for tag, kind, value in tags:
            fields.append("%s:%s:%s" % (tag, kind, "#%08x" % struct.unpack("<I", struct.pack("<f", value))[0] if kind == "f" else value))
        lines.append("\t".join(fields))
        expected.append((name, flag, sequence.upper(), quality, tags))
    header = "@HD\tVN:1.0\tSO:unknown\tGO:query\n@RG\tID:group:1\tPU:x\n"
    (tmp_path / "lines.sam").write_text("\n".join(lines) + "\n")
    (tmp_path / "header.sam").write_text(header)
    ran = subprocess.run([check, str(tmp_path / "lines.sam"), str(tmp_path / "out.bam"), str(tmp_path / "header.sam")], capture_output=True, text=True)
    assert ran.returncode == 0, ran.stdout + ran.stderr
    text, records, blocks = bam_reader.records(str(tmp_path / "out.bam"))
    assert text == header and len(records) == len(expected) and blocks > 10
    for record, (name, flag, sequence, quality, tags) in zip(records, expected):
        assert (record["name"], record["flag"], record["sequence"], record["quality"]) == (name, flag, sequence, quality)
        assert (record["ref"], record["pos"], record["mapq"], record["bin"], record["n_cigar"], record["next_ref"], record["next_pos"], record["tlen"]) == (-1, -1, 0, 0, 0, -1, -1, 0)
        assert record["tags"] == tags                   # (the floats bit for bit: both sides went through the same four bytes)
    # no lines: a header and the end of file block
    (tmp_path / "none.sam").write_text("")
    assert subprocess.run([check, str(tmp_path / "none.sam"), str(tmp_path / "none.bam")], capture_output=True).returncode == 0
    assert bam_reader.records(str(tmp_path / "none.bam"))[:2] == ("", [])
    # lines this writer does not make
    good = lines[1]
    for bad, message in (("\t".join(good.split("\t")[:9]), "fewer than eleven fields"),
                         ("q" * 255 + good[good.index("\t"):], "254"),
                         (good + "\tXB:f:0.25", "float tag"),
                         (good + "\tXB:f:#0000zz00", "states no bits"),
                         (good + "\tXY:A:c", "tag type"),
                         (good + "\tbroken", "tag this writer does not make")):
        (tmp_path / "bad.sam").write_text(bad + "\n")
        ran = subprocess.run([check, str(tmp_path / "bad.sam"), str(tmp_path / "bad.bam")], capture_output=True, text=True)
        assert ran.returncode == 3 and message in ran.stderr, (bad[-40:], ran.returncode, ran.stderr[-300:])
