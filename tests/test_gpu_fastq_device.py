"""Report only runs decode their FASTQ text on the device (csrc/fastq_device.hip): line ends, framing, name sync, the
Illumina filter flag and the decoders' tokens (fastq.h:55-78,253-285; read.h:253-264; transform.h:65-80,142-169).  The
counters such a run leaves are held against the CPU ORACLE's for the same file content (its own Rule::apply over the reads'
text, `~` tokens included), and word for word against the host reader's (PHQ_HOST_FASTQ=1); anything out of the ordinary
in a file sends the run through the host reader, which tolerates or reports it as the reference does.
"""
import json
import os

import numpy as np
import pytest

import bdggg
import bench
from pheniqs_amd import DecoderSet, PheniqsError, capi, synthetic

pytestmark = pytest.mark.gpu

LETTERS = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)


COMPLEMENT = np.arange(16, dtype=np.uint8)
COMPLEMENT[[1, 2, 4, 8]] = [8, 4, 2, 1]                     # iupac.h:107-124 on the unambiguous codes; N stays N
INTEGER_COUNTERS = ("count", "pf_count", "accumulated_distance", "accumulated_pf_distance",
                    "low_conditional_confidence_count", "low_confidence_count")


def _files(directory, spec, n, seed, lengths, content=None):
    """FASTQ files of a job: every decoder's synthetic observation written where its tokens read it; a reverse complemented
    token (`~`) gets the reverse complement of the observation, so that Rule::apply gives the observation back.  `content`,
    a list, receives (text, quality, flag) of every file as written."""
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, seed)
    rng = np.random.default_rng(seed)
    text = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, length), dtype=np.uint8)] for length in lengths]
    qual = [np.full((n, length), ord("F"), dtype=np.uint8) for length in lengths]
    for (_, decoder), offset in zip(synthetic.iter_decoders(job), offsets):
        at = offset
        for token in decoder["transform"]["token"]:
            segment, start, end = token.split(":")
            segment, start = int(segment), int(start or 0)
            width = (int(end) if end else lengths[segment]) - start
            knit = decoder["transform"].get("knit")
            piece, piece_quality = code[:, at:at + width], np.minimum(quality[:, at:at + width], 93)
            if knit and knit[0].startswith("~"):
                assert len(decoder["transform"]["token"]) == 1
                piece, piece_quality = COMPLEMENT[piece[:, ::-1]], piece_quality[:, ::-1]
            text[segment][:, start:start + width] = LETTERS[piece]
            qual[segment][:, start:start + width] = piece_quality + 33
            at += width
    paths = []
    for k in range(len(lengths)):
        path = os.path.join(directory, "segment_%d.fastq" % k)
        bench.write_fastq(path, k + 1, text[k], qual[k], flag)
        paths.append(path)
    if content is not None:
        content.extend([text, qual, flag])
    return paths


def _assert_counters_equal_the_oracle(oracle, spec, content, words):
    """The oracle decodes the files' content by itself — its own ASCII to BAM table, its own Rule::apply (orc_stage_read, `~`
    included), its own classify — and the device run's counter words must say the same, decoder by decoder."""
    text, qual, flag = content
    job = capi.compile_job(spec)
    engine = oracle.OracleJob(job)
    codes = [oracle.ascii_to_bam(t.tobytes()).reshape(t.shape) for t in text]
    qualities = [(q - 33).astype(np.uint8) for q in qual]
    n = flag.shape[0]
    base, quality = engine.stage([[(codes[k][i], qualities[k][i]) for k in range(len(text))] for i in range(n)])
    engine.classify(base, quality, flag)
    s = DecoderSet(spec, device=0)
    count, pf_count = engine.outgoing()
    assert (int(words[0] + words[1]), int(words[1])) == (count, pf_count)
    for k in range(s.n_decoder):
        mine, theirs = s.stats_options(words, k), engine.options(k)
        for name in INTEGER_COUNTERS:
            assert np.array_equal(mine[name], theirs[name]), (k, name)
        for name in ("accumulated_confidence", "accumulated_pf_confidence"):
            assert np.allclose(mine[name], theirs[name], rtol=1e-8, atol=1e-8), (k, name)
    s.close()


def _both(spec, paths, capacity):
    """(counter words, incoming) of the device path and of the host reader on the same files"""
    out = []
    for host in (False, True):
        if host:
            os.environ["PHQ_HOST_FASTQ"] = "1"
        else:
            os.environ.pop("PHQ_HOST_FASTQ", None)
        try:
            s = DecoderSet(spec, device=0)
            s.stats_reset()
            incoming = s.transcode_fastq_to_sam(paths, "/dev/null", capacity)
            out.append((s.stats_export(), incoming))
            s.close()
        finally:
            os.environ.pop("PHQ_HOST_FASTQ", None)
    return out


@pytest.mark.parametrize("config,lengths,n,capacity,zipped", [("C2", (50, 8), 150000, 1 << 20, False), ("C2", (50, 8), 150000, 1 << 12, False),
                                                              ("C5", (20, 100), 60000, 1 << 14, False), ("C4", (30, 10, 10), 60000, 1 << 20, False),
                                                              ("C2", (50, 8), 150000, 1 << 12, True), ("C4", (30, 10, 10), 60000, 1 << 13, True),
                                                              ("C2", (50, 8), 150000, 1 << 14, "bgzf"), ("C4", (30, 10, 10), 60000, 1 << 12, "members")])
def test_device_decode_equals_the_oracle_and_the_host_reader(oracle, tmp_path, config, lengths, n, capacity, zipped):
    """C5's fourth round is a `~` token: the files hold the reverse complement of its observation; `zipped`: every file a gzip
    stream, inflated by a host thread per file into the pinned buffer the text is uploaded from; "bgzf" / "members": bgzip's
    blocks (what the reference reads FASTQ through, fastq.h:427-440) and plain concatenated members of uneven sizes — inflated
    member parallel, a batch ahead of the rounds (csrc/gzip_members.hpp), the first file only of the "members" case (the others
    stay single streams: both kinds side by side in one run)"""
    spec = synthetic.config_spec(config, "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    content = []
    paths = _files(str(tmp_path), spec, n, 77, lengths, content)
    if zipped:
        import gzip
        from test_capi_cpu import _bgzf
        for k, path in enumerate(paths):
            data = open(path, "rb").read()
            if zipped == "bgzf":
                packed = _bgzf(data)
            elif zipped == "members" and k == 0:
                cuts = [0, 777, len(data) // 5, len(data) // 2, len(data) - 1, len(data)]                 # cut anywhere: mid record, mid line
                packed = b"".join(gzip.compress(data[a:b], 1 + (a % 9)) for a, b in zip(cuts[:-1], cuts[1:]))
            else:
                packed = gzip.compress(data, 1)
            open(path + ".gz", "wb").write(packed)
            paths[k] = path + ".gz"
        before = capi.device_path_counters()
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, capacity)
    assert device_incoming == host_incoming and device_incoming[0] == n
    assert int(device_words[0] + device_words[1]) == n
    assert np.array_equal(device_words, host_words)
    _assert_counters_equal_the_oracle(oracle, spec, content, device_words)
    if zipped:
        after = capi.device_path_counters()
        assert after["reads"] - before["reads"] == n and after["rounds"] - before["rounds"] == -(-n // max(capacity, 4096))


@pytest.mark.parametrize("packing", ["gz", "bgzf"])
def test_gzip_rounds_that_take_little_of_their_text(oracle, tmp_path, packing):
    """A gzip feed is inflated a round AHEAD, behind room for what the round before leaves (an eighth of a round's bytes).  A file
    whose records get SHORTER than its first ones said — 150 nt templates at its head, 20 nt ones behind — has rounds that take
    their capacity of reads out of a quarter of their bytes: what was inflated ahead is set aside and follows when there is room
    again.  Counters against the oracle and the host reader, the order of the reads with them (a read out of place would be
    another read's index)."""
    import gzip
    from test_capi_cpu import _bgzf
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    n_long, n_short = 3000, 90000
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, n_long + n_short, 5)
    rng = np.random.default_rng(5)
    pieces = [[], []]
    for first, last, length in ((0, n_long, 150), (n_long, n_long + n_short, 20)):
        rows = last - first
        template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(rows, length), dtype=np.uint8)]
        index = LETTERS[code[first:last]]
        for k, (text, qual) in enumerate(((template, np.full((rows, length), ord("F"), dtype=np.uint8)), (index, np.minimum(quality[first:last], 93) + 33))):
            path = os.path.join(str(tmp_path), "piece.fastq")
            bench.write_fastq(path, k + 1, text, qual.astype(np.uint8), flag[first:last])
            pieces[k].append(open(path, "rb").read())
    paths = []
    for k in range(2):
        data = b"".join(pieces[k])
        paths.append(os.path.join(str(tmp_path), "segment_%d.fastq.gz" % k))
        open(paths[-1], "wb").write(_bgzf(data) if packing == "bgzf" else gzip.compress(data, 4))
    before = capi.device_path_counters()
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, 4096)
    after = capi.device_path_counters()
    assert after["reads"] - before["reads"] == n_long + n_short and after["resumed"] == before["resumed"]
    assert device_incoming == host_incoming and device_incoming[0] == n_long + n_short
    assert np.array_equal(device_words, host_words)
    engine = oracle.OracleJob(job)
    engine.classify(code, quality, flag)
    s = DecoderSet(spec, device=0)
    mine, theirs = s.stats_options(device_words, 0), engine.options(0)
    for name in INTEGER_COUNTERS:
        assert np.array_equal(mine[name], theirs[name]), name
    s.close()


def test_reverse_complement_tokens_and_negative_offsets_on_the_device(oracle, tmp_path):
    """tokens the forward jobs above do not hold: `~` on a two token knit, an offset from the end of the segment, an open
    start (transform.h:65-80,142-169) — the oracle's Rule::apply over the same text is the reference"""
    rng = np.random.default_rng(5)
    words = synthetic.barcode_set(rng, 24, 8)
    n, lengths = 40000, (30, 12)
    spec = {"input segment cardinality": 2, "template": {"transform": {"token": ["0::"]}},
            "sample": synthetic.decoder_spec("pamld", [words], ["1:2:6", "1:-6:-2"], knit=["~1:~0"]),
            "molecular": [{"algorithm": "naive", "transform": {"token": ["0:-5:"], "knit": ["~0"]}}]}
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 11)
    text = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, length), dtype=np.uint8)] for length in lengths]
    qual = [np.full((n, length), ord("F"), dtype=np.uint8) for length in lengths]
    # observation = rc(token 1) + rc(token 0): token 1 = segment 1 [6, 10) holds rc of the observation's first half,
    # token 0 = segment 1 [2, 6) holds rc of its second half
    observed, observed_quality = code[:, :8], np.minimum(quality[:, :8], 93)
    for at, half in ((6, slice(0, 4)), (2, slice(4, 8))):
        text[1][:, at:at + 4] = LETTERS[COMPLEMENT[observed[:, half][:, ::-1]]]
        qual[1][:, at:at + 4] = observed_quality[:, half][:, ::-1] + 33
    paths = []
    for k in range(2):
        path = os.path.join(str(tmp_path), "segment_%d.fastq" % k)
        bench.write_fastq(path, k + 1, text[k], qual[k], flag)
        paths.append(path)
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, 1 << 14)
    assert device_incoming == host_incoming and device_incoming[0] == n
    assert np.array_equal(device_words, host_words)
    _assert_counters_equal_the_oracle(oracle, spec, [text, qual, flag], device_words)
    s = DecoderSet(spec, device=0)
    classified = int(s.stats_options(device_words, 0)["count"][1:].sum())
    s.close()
    assert classified > 0.9 * n                       # the reads do decode: the tokens cut what the files hold


def test_oddities_go_through_the_host_reader(tmp_path):
    """no newline at the end, carriage returns, a blank line at the end: the device path hands the run over and the host
    reader's result stands; names out of sync and a truncated file raise the host reader's errors"""
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    paths = _files(str(tmp_path), spec, 5000, 5, (50, 8))
    reference = _both(spec, paths, 1 << 20)
    assert np.array_equal(reference[0][0], reference[1][0])
    original = [open(p, "rb").read() for p in paths]
    variants = {"no last newline": [original[0][:-1], original[1][:-1]],
                "carriage returns": [original[0].replace(b"\n", b"\r\n"), original[1]],
                "blank line at the end": [original[0] + b"\n", original[1] + b"\n"]}
    for name, pair in variants.items():
        for path, text in zip(paths, pair):
            open(path, "wb").write(text)
        (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, 1 << 20)
        assert device_incoming == host_incoming, name
        assert np.array_equal(device_words, host_words), name
    # a name that differs in one file, a file cut inside a record, a file one record short
    broken = {"out of sync": [original[0].replace(b":0000000:00017 ", b":0000000:99999 ", 1), original[1]],
              "truncated": [original[0][:len(original[0]) // 2 + 7], original[1]],
              "one record short": [original[0], original[1][:original[1].rfind(b"@M:1")]]}
    assert broken["out of sync"][0] != original[0]
    for name, pair in broken.items():
        for path, text in zip(paths, pair):
            open(path, "wb").write(text)
        codes = []
        for host in (False, True):
            if host:
                os.environ["PHQ_HOST_FASTQ"] = "1"
            try:
                s = DecoderSet(spec, device=0)
                with pytest.raises(PheniqsError) as raised:
                    s.transcode_fastq_to_sam(paths, "/dev/null", 1 << 20)
                codes.append(raised.value.code)
                assert int(s.stats_export().sum()) == 0 or host, name          # the device path counted nothing it keeps
                s.close()
            finally:
                os.environ.pop("PHQ_HOST_FASTQ", None)
        assert codes[0] == codes[1], name


def _device_path_taken(job_json, run):
    """Runs `run(set)` twice — device FASTQ decode, then the host reader — and returns both (incoming, counter words, report)."""
    out = []
    for host in (False, True):
        if host:
            os.environ["PHQ_HOST_FASTQ"] = "1"
        try:
            s = DecoderSet(job_json, device=0)
            s.stats_reset()
            incoming = run(s)
            words = s.stats_export()
            out.append((incoming, words, json.loads(s.report(words, incoming))))
            s.close()
        finally:
            os.environ.pop("PHQ_HOST_FASTQ", None)
    return out


def _assert_report_is_the_golden_one(report):
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


def test_the_reference_job_report_only_on_the_device(tmp_path):
    """test/BDGGG's own job (BDGGG_annotated.json: `min input length` [0, 0, 22], three decoders) with every output /dev/null
    runs its FASTQ text through the device decode — the pre-filter of the pull loop (transcode.h:193-205) included — and its
    report is valid/annotated.err's; the same files as ONE interleaved feed (transcode.cpp:693-701) likewise."""
    job = bdggg.raw_job()
    job["base input url"] = bdggg.HERE
    job["output"] = ["/dev/null"]
    device, host = _device_path_taken(json.dumps(job), lambda s: s.transcode_job(100))
    assert device[0] == host[0] == (250, 243)
    assert np.array_equal(device[1], host[1])
    _assert_report_is_the_golden_one(device[2])
    # interleaved: record 3 i + k of the one file is segment k of read i
    sources = [open(os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k)).read().split("\n") for k in (1, 2, 3)]
    n = min(len(lines) // 4 for lines in sources)
    path = os.path.join(str(tmp_path), "interleaved.fastq")
    with open(path, "w") as out:
        for i in range(n):
            for lines in sources:
                out.write("\n".join(lines[4 * i:4 * i + 4]) + "\n")
    job["input"] = [path, path, path]
    device, host = _device_path_taken(json.dumps(job), lambda s: s.transcode_job(64))
    assert device[0] == host[0] == (250, 243)
    assert np.array_equal(device[1], host[1])
    _assert_report_is_the_golden_one(device[2])


def test_pre_filters_compact_the_reads_on_the_device(oracle, tmp_path):
    """`filter incoming qc fail` and `min input length` on the device (transcode.h:193-205): reads the pull loop drops are
    counted as incoming and nothing else; what is left must count exactly as the oracle counts the kept reads, in several
    rounds (a capacity far under the file's reads) — the device, the host reader and the oracle agree word for word."""
    rng = np.random.default_rng(9)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    spec["filter incoming qc fail"] = True
    spec["min input length"] = [0, 6]
    job = capi.compile_job(spec)
    n = 90001
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 31)
    template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 30))]
    short = (np.arange(n) % 89) == 0                          # an index read of 5 nt: under the minimum of 6
    paths = [os.path.join(str(tmp_path), "s%d.fastq" % k) for k in (0, 1)]
    with open(paths[0], "wb") as a, open(paths[1], "wb") as b:
        for i in range(n):
            filtered = b"Y" if flag[i] else b"N"
            keep = 5 if short[i] else 8
            a.write(b"@r%d 1:%s:0:1\n%s\n+\n%s\n" % (i, filtered, template[i].tobytes(), b"F" * 30))
            b.write(b"@r%d 2:%s:0:1\n%s\n+\n%s\n" % (i, filtered, LETTERS[code[i, :keep]].tobytes(), (np.minimum(quality[i, :keep], 93) + 33).astype(np.uint8).tobytes()))
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, 1 << 13)
    assert device_incoming == host_incoming == (n, n - int(flag.sum()))
    assert np.array_equal(device_words, host_words)
    kept = (flag == 0) & ~short
    engine = oracle.OracleJob(job)
    engine.classify(code[kept], np.minimum(quality[kept], 93), np.zeros(int(kept.sum()), dtype=np.uint8))
    count, pf_count = engine.outgoing()
    assert (int(device_words[0] + device_words[1]), int(device_words[1])) == (count, pf_count)
    assert count == int(kept.sum())
    s = DecoderSet(spec, device=0)
    mine, theirs = s.stats_options(device_words, 0), engine.options(0)
    for name in INTEGER_COUNTERS:
        assert np.array_equal(mine[name], theirs[name]), name
    s.close()
