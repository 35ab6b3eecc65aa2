"""Report only runs decode their FASTQ text on the device (csrc/fastq_device.hip): line ends, framing, name sync, the
Illumina filter flag and the decoders' tokens (fastq.h:55-78,253-285; read.h:253-264; transform.h:65-80,142-169).  The
counters such a run leaves are held against the CPU ORACLE's for the same file content (its own Rule::apply over the reads'
text, `~` tokens included), and word for word against the host reader's (PHQ_HOST_FASTQ=1); anything out of the ordinary
in a file sends the run through the host reader, which tolerates or reports it as the reference does.
"""
import os

import numpy as np
import pytest

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


@pytest.mark.parametrize("config,lengths,n,capacity", [("C2", (50, 8), 150000, 1 << 20), ("C2", (50, 8), 150000, 1 << 12),
                                                       ("C5", (20, 100), 60000, 1 << 14), ("C4", (30, 10, 10), 60000, 1 << 20)])
def test_device_decode_equals_the_oracle_and_the_host_reader(oracle, tmp_path, config, lengths, n, capacity):
    """C5's fourth round is a `~` token: the files hold the reverse complement of its observation"""
    spec = synthetic.config_spec(config, "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    content = []
    paths = _files(str(tmp_path), spec, n, 77, lengths, content)
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, capacity)
    assert device_incoming == host_incoming and device_incoming[0] == n
    assert int(device_words[0] + device_words[1]) == n
    assert np.array_equal(device_words, host_words)
    _assert_counters_equal_the_oracle(oracle, spec, content, device_words)


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
