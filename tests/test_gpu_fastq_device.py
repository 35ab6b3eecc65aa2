"""Report only runs decode their FASTQ text on the device (csrc/fastq_device.hip): line ends, framing, name sync, the
Illumina filter flag and the decoders' tokens (fastq.h:55-78,253-285; read.h:253-264; transform.h:65-80,142-169).  The
same files through the host reader (PHQ_HOST_FASTQ=1) must leave the same counters word for word; anything out of the
ordinary in a file sends the run through the host reader, which tolerates or reports it as the reference does.
"""
import os

import numpy as np
import pytest

import bench
from pheniqs_amd import DecoderSet, PheniqsError, capi, synthetic

pytestmark = pytest.mark.gpu

LETTERS = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)


def _files(directory, spec, n, seed, lengths):
    """FASTQ files of a job: every decoder's synthetic observation written where its tokens read it (forward tokens)."""
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
            if knit and knit[0].startswith("~"):
                continue                                  # a reverse complemented observation: left as random text
            text[segment][:, start:start + width] = LETTERS[code[:, at:at + width]]
            qual[segment][:, start:start + width] = np.minimum(quality[:, at:at + width], 93) + 33
            at += width
    paths = []
    for k in range(len(lengths)):
        path = os.path.join(directory, "segment_%d.fastq" % k)
        bench.write_fastq(path, k + 1, text[k], qual[k], flag)
        paths.append(path)
    return paths


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
def test_device_decode_equals_host_reader(tmp_path, config, lengths, n, capacity):
    spec = synthetic.config_spec(config, "pamld")
    spec["template"] = {"transform": {"token": ["0::"]}}
    paths = _files(str(tmp_path), spec, n, 77, lengths)
    (device_words, device_incoming), (host_words, host_incoming) = _both(spec, paths, capacity)
    assert device_incoming == host_incoming and device_incoming[0] == n
    assert int(device_words[0] + device_words[1]) == n
    assert np.array_equal(device_words, host_words)


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
