"""Output records assembled on the device (csrc/record_device.hip): a job whose every written channel goes to one
uncompressed SAM or FASTQ file has its records put together by kernels — Read::flush and Auxiliary::encode (read.h:187-237,
auxiliary.cpp:324-354), append_corrected (sequence.h:382-398), TemplateRule::apply (transform.h:190-226), FastqRecord::encode
(fastq.h:144-194), Channel::push (multiplex.h:198-240).  The reference's own golden SAM file is the first check (the BDGGG
job; tests/test_gpu_parity.py holds the same file against every other route); the host writer — itself held against that
file — is the second, byte for byte, on jobs the golden file has no case of: corrected barcodes as template tokens, `~`,
offsets from the end, more than two output segments, FASTQ output, the outgoing filter, pre-filters, many rounds.  An input
that turns out of the ordinary part way is handed to the host reader at the byte the device stopped at.
"""
import json
import os

import numpy as np
import pytest

import bdggg
from pheniqs_amd import DecoderSet, PheniqsError, capi, synthetic
from test_gpu_fastq_device import _files

pytestmark = pytest.mark.gpu


def _run(job, capacity, host, how="job", paths=None, output=None, mode=None):
    """(incoming, counter words, device path counters of the run) with the records assembled on the device or by the host"""
    if host:
        os.environ["PHQ_HOST_RECORDS"] = "1"
    try:
        before = capi.device_path_counters()
        s = DecoderSet(job if isinstance(job, str) else json.dumps(job), device=0)
        if mode is not None:
            s.set_mode(mode)
        s.stats_reset()
        incoming = s.transcode_job(capacity) if how == "job" else s.transcode_fastq_to_sam(paths, output, capacity)
        words = s.stats_export()
        did = {key: value - before[key] for key, value in capi.device_path_counters().items()}
        did["quality_control"] = s.qc_export()
        s.close()
        return incoming, words, did
    finally:
        os.environ.pop("PHQ_HOST_RECORDS", None)


def _records(path):
    """the file without the @PG line (it names the program)"""
    return b"".join(line for line in open(path, "rb").readlines() if not line.startswith(b"@PG"))


def test_the_golden_file_from_the_device(tmp_path):
    """valid/annotated.out byte for byte with every record assembled on the device: three input files, and the same reads
    interleaved in one; one round each."""
    golden = open(os.path.join(bdggg.HERE, "annotated.out.sam"), "rb").read()
    job = bdggg.raw_job()
    job["base input url"] = bdggg.HERE
    job["base output url"] = str(tmp_path)
    job["output"] = ["out.sam"]
    incoming, _, did = _run(job, 100, host=False)
    assert incoming == (250, 243)
    assert did["rounds"] == 1 and did["reads"] == 250 and did["resumed"] == 0
    mine = _records(os.path.join(str(tmp_path), "out.sam"))
    assert mine == golden
    assert did["record_bytes"] == sum(len(line) for line in golden.split(b"\n") if line and not line.startswith(b"@")) + 496
    sources = [open(os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k)).read().split("\n") for k in (1, 2, 3)]
    path = os.path.join(str(tmp_path), "interleaved.fastq")
    with open(path, "w") as out:
        for i in range(250):
            for lines in sources:
                out.write("\n".join(lines[4 * i:4 * i + 4]) + "\n")
    job["input"] = [path, path, path]
    job["output"] = ["again.sam"]
    incoming, _, did = _run(job, 64, host=False)
    assert incoming == (250, 243) and did["rounds"] == 1 and did["record_bytes"] > 0
    assert _records(os.path.join(str(tmp_path), "again.sam")) == golden


def _mixed_spec():
    """every kind of decoder and tag: a PAMLD sample barcode, a naive and an MDD molecular barcode (RX QX OX BZ XM), a PAMLD and
    a two segment MDD cellular barcode (CB CR CY XC); the template takes a slice from the end, the corrected sample, cellular
    and molecular barcodes as tokens, `~` on two of them, and makes four output segments (FI, TC)"""
    rng = np.random.default_rng(404)
    sample = synthetic.barcode_set(rng, 24, 8)
    umi = synthetic.barcode_set(rng, 16, 6)
    a, b, c = synthetic.barcode_set(rng, 24, 10), synthetic.barcode_set(rng, 12, 10), synthetic.barcode_set(rng, 12, 10)
    return {"input segment cardinality": 3,
            "sample": synthetic.decoder_spec("pamld", [sample], ["1::8"]),
            "molecular": [{"algorithm": "naive", "transform": {"token": ["0::8"]}}, synthetic.decoder_spec("mdd", [umi], ["0:8:14"])],
            "cellular": [synthetic.decoder_spec("pamld", [a], ["0:14:24"]), synthetic.decoder_spec("mdd", [b, c], ["2::10", "2:10:20"], pairing="zip")],
            "template": {"transform": {"token": ["0:24:", "s::", "c::", "m:1:", "2:-6:", "1::4", "0:24:30"], "knit": ["0", "~1:2", "3:~4", "5:~6"]}}}


@pytest.mark.parametrize("case", ["sam_two_segments_filters", "sam_mixed_tokens", "fastq_interleaved", "outgoing_filter", "one_channel_silent",
                                  "long_reads", "lengths_change", "gzip_input"])
def test_device_records_equal_the_host_writers(tmp_path, case):
    directory = str(tmp_path)
    if case in ("sam_two_segments_filters", "fastq_interleaved", "outgoing_filter", "one_channel_silent", "long_reads", "lengths_change", "gzip_input"):
        spec = synthetic.config_spec("C2", "pamld")
        spec["template"] = {"transform": {"token": ["0::", "1::"]}}
        lengths, n, capacity = (30, 8), 70001, 4096
        if case == "long_reads":
            lengths, n = (301, 8), 20011                    # 16 reads to a wave: their text is 12 KiB
        if case == "sam_two_segments_filters":
            spec["filter incoming qc fail"] = True
            spec["min input length"] = [0, 6]
        if case == "outgoing_filter":
            spec["filter outgoing qc fail"] = True
    else:
        spec = _mixed_spec()
        lengths, n, capacity = (40, 8, 26), 30011, 1 << 13
    paths = _files(directory, spec, n, 5, lengths)
    if case == "lengths_change":
        # the first 64 KiB of a file size the waves' rooms: after 3 000 reads of 30 nt come reads of 700 nt whose text fits
        # neither the room of a wave's input nor that of its output — both then go straight to and from memory
        os.makedirs(os.path.join(directory, "long"))
        longer = _files(os.path.join(directory, "long"), spec, 9000, 6, (700, 8))
        for short, long in zip(paths, longer):
            lines = open(short, "rb").read().split(b"\n")[:4 * 3000]
            open(short, "wb").write(b"\n".join(lines) + b"\n" + open(long, "rb").read())
        n = 12000
    if case == "gzip_input":
        # the template file as a gzip stream of two members, the index file plain: the stream is inflated into the pinned buffer
        import gzip
        data = open(paths[0], "rb").read()
        cut = data.index(b"\n@", len(data) // 2) + 1
        with open(paths[0] + ".gz", "wb") as out:
            out.write(gzip.compress(data[:cut], 1))
            out.write(gzip.compress(data[cut:], 1))
        paths[0] += ".gz"
    if case == "sam_two_segments_filters":
        # a few index reads under the minimum length: the pull loop drops those reads
        lines = open(paths[1], "rb").read().split(b"\n")
        for i in range(0, n, 97):
            lines[4 * i + 1] = lines[4 * i + 1][:5]
            lines[4 * i + 3] = lines[4 * i + 3][:5]
        open(paths[1], "wb").write(b"\n".join(lines))
    spec["input"] = paths
    outputs = {}
    for host in (False, True):
        name = ("host" if host else "device") + (".fastq" if case == "fastq_interleaved" else ".sam")
        spec["output"] = [os.path.join(directory, name)]
        if case == "one_channel_silent":
            first = next(iter(spec["sample"]["codec"]))
            spec["sample"]["codec"][first]["output"] = ["/dev/null"]
        incoming, words, did = _run(spec, capacity, host)
        outputs[host] = (incoming, words, _records(os.path.join(directory, name)), did)
    device, host = outputs[False], outputs[True]
    assert device[0] == host[0] and device[0][0] == n
    assert np.array_equal(device[1], host[1])
    # (a round takes as many reads as the bytes sized for `capacity` reads hold: more rounds where the reads grow longer)
    assert device[3]["record_bytes"] > 0 and device[3]["resumed"] == 0
    assert device[3]["rounds"] >= -(-n // capacity) if case == "lengths_change" else device[3]["rounds"] == -(-n // capacity)
    assert host[3]["record_bytes"] == 0
    assert len(device[2]) == len(host[2])
    if device[2] != host[2]:
        mine, theirs = device[2].split(b"\n"), host[2].split(b"\n")
        for k, (a, b) in enumerate(zip(mine, theirs)):
            assert a == b, (k, a, b)
    assert device[2] == host[2]
    body = b"".join(line + b"\n" for line in device[2].split(b"\n") if line and not line.startswith(b"@"))
    if case != "fastq_interleaved":
        assert device[3]["record_bytes"] == len(body)
    if case == "sam_mixed_tokens":
        text = device[2].decode()
        for tag in ("\tFI:i:4\tTC:i:4", "\tRG:Z:", "\tBC:Z:", "\tQT:Z:", "\tXB:f:", "\tRX:Z:", "\tQX:Z:", "\tOX:Z:", "\tBZ:Z:", "\tCB:Z:", "\tCR:Z:", "\tCY:Z:", "\tXC:f:"):
            assert tag in text, tag
    if case == "outgoing_filter":
        assert b"\t589\t" not in device[2] and b"\t77\t" in device[2]            # no record of a read that failed QC
    if case == "one_channel_silent":
        assert 0 < device[2].count(b"\n") < 2 * n                                # the silent channel's reads leave no record


def _assert_sam_equal_but_for_posterior_digits(mine, expected):
    """SAM records byte for byte, but for the digits of XB / XM / XC (see test_records_equal_the_independent_python_restatement)"""
    mine_lines, expected_lines = mine.split(b"\n"), expected.split(b"\n")
    assert len(mine_lines) == len(expected_lines)
    identical = 0
    for k, (a, b) in enumerate(zip(mine_lines, expected_lines)):
        if a == b:
            identical += 1
            continue
        fa, fb = a.split(b"\t"), b.split(b"\t")
        assert len(fa) == len(fb), (k, a, b)
        for x, y in zip(fa, fb):
            if x != y:
                assert x[:5] == y[:5] and x[:5] in (b"XB:f:", b"XM:f:", b"XC:f:"), (k, a, b)
                u, v = float(x[5:]), float(y[5:])
                assert x[5:] == b"%g" % u and abs(u - v) <= 2e-8 + 1e-6 * v, (k, a, b)
    assert identical >= 0.5 * len(mine_lines)


@pytest.mark.parametrize("case", ["two_segments_filters", "mixed_tokens", "three_output_segments", "fastq_interleaved", "fastq_mixed_tokens"])
@pytest.mark.parametrize("host", [False, True])
@pytest.mark.parametrize("mode", ["auto", "fast"])
def test_records_equal_the_independent_python_restatement(oracle, tmp_path, case, host, mode):
    """tests/sam_restatement.py — written from the reference's text and pinned to its golden SAM file without the library
    (tests/test_sam_restatement.py) — formats the ORACLE's decisions for the same files; the device writer and the host writer
    must each write the same bytes: tags in Auxiliary::encode's order, `%g`, FI / TC, corrected barcodes as template tokens with
    the S9 probe, `~`, offsets from the end, the incoming filters.  Not a comparison of the library with itself."""
    import sam_restatement
    directory = str(tmp_path)
    fastq = case.startswith("fastq")
    if case in ("mixed_tokens", "fastq_mixed_tokens"):
        spec = _mixed_spec()
        lengths, n, capacity = (40, 8, 26), 6007, 1 << 12
    else:
        spec = synthetic.config_spec("C2", "pamld")
        spec["template"] = {"transform": {"token": ["0::", "1::"]}}
        lengths, n, capacity = (30, 8), 8009, 4096
        if case == "two_segments_filters":
            spec["filter incoming qc fail"] = True
            spec["min input length"] = [0, 6]
        elif case == "three_output_segments":
            spec["template"] = {"transform": {"token": ["0::12", "0:-9:", "1::", "s::"], "knit": ["0", "~2:3", "1"]}}
    paths = _files(directory, spec, n, 9, lengths)
    if case == "two_segments_filters":
        lines = open(paths[1], "rb").read().split(b"\n")
        for i in range(0, n, 97):
            lines[4 * i + 1] = lines[4 * i + 1][:5]
            lines[4 * i + 3] = lines[4 * i + 3][:5]
        open(paths[1], "wb").write(b"\n".join(lines))
    spec["input"] = paths
    spec["output"] = [os.path.join(directory, "out.fastq" if fastq else "out.sam")]
    incoming, _, did = _run(spec, capacity, host, mode=capi.MODE_FAST if mode == "fast" else None)
    assert (did["record_bytes"] == 0) == host
    mine = open(spec["output"][0], "rb").read() if fastq else b"".join(line for line in open(spec["output"][0], "rb").readlines() if not line.startswith(b"@"))

    job = capi.compile_job(spec)
    files = [sam_restatement.read_fastq(path) for path in paths]
    minimum = job.get("min input length") or []
    kept = [k for k in range(n) if not ((job.get("filter incoming qc fail") and files[0][k][1]) or
                                        any(len(files[i][k][2]) < minimum[i] for i in range(1, min(len(minimum), len(files)))))]
    engine = oracle.OracleJob(job)
    base, quality = engine.stage([[(files[i][k][2], files[i][k][3]) for i in range(len(files))] for k in kept])
    flags = np.array([1 if files[0][k][1] else 0 for k in kept], dtype=np.uint8)
    result, word, _, confidence = engine.classify(base, quality, flags, summary=True)
    expected = (sam_restatement.fastq_records if fastq else sam_restatement.records)(job, files, result, word, confidence)
    assert incoming[0] == n and len(kept) < n or case != "two_segments_filters"
    # byte for byte, but for the digits of XB / XM / XC: the tag prints float(1 - posterior) with six digits, and the fast mode's
    # posterior is the oracle's to 1e-8 (tests/test_gpu_parity.py), not to the bit — 1e-8 of the posterior is the fourth digit of
    # a 1e-5 tag.  Those tags must parse, sit where the oracle's sit and agree to what 1e-8 leaves; most lines are identical
    if fastq:
        # (a FASTQ record carries no posterior: byte for byte)
        if mine != expected:
            for k, (a, b) in enumerate(zip(mine.split(b"\n"), expected.split(b"\n"))):
                assert a == b, (k, a, b)
        assert mine == expected
        assert (b" 4:N:0:" in mine or b" 4:Y:0:" in mine) == (case == "fastq_mixed_tokens")
        return
    cells = sum(len(d["codec"]) * int(d["nucleotide cardinality"]) for _, d in synthetic.iter_decoders(job) if d.get("algorithm") == "pamld")
    if mode == "auto" and cells <= 4096:
        # PHQ_MODE_AUTO, the default: a run that writes SAM records of a job this small classifies in the literal kernel's
        # arithmetic — the posterior is the oracle's to ~1e-16 and the tags' six digits are the reference's: byte for byte, no exemption
        if mine != expected:
            for k, (a, b) in enumerate(zip(mine.split(b"\n"), expected.split(b"\n"))):
                assert a == b, (k, a, b)
        assert mine == expected
    else:
        _assert_sam_equal_but_for_posterior_digits(mine, expected)
    if case == "mixed_tokens":
        assert b"\tFI:i:4\tTC:i:4" in mine and b"\tXC:f:" in mine and b"\tRX:Z:" in mine
    if case == "three_output_segments":
        assert b"\tFI:i:3\tTC:i:3" in mine


@pytest.mark.parametrize("case", ["C2", "mixed_tokens", "per_barcode"])
def test_unaligned_bam_output_holds_the_sam_records(tmp_path, case):
    """`output: x.bam` (what most Pheniqs runs state; the reference writes it through htslib, hts.h:169-239): the library writes
    unaligned BAM itself — records made from the SAM lines its writers format, on the device or on the host (csrc/bam_record.hpp),
    BGZF blocks by its own deflate, the empty block at the end.  tests/bam_reader.py reads the files back from the specification alone: every BGZF block
    states its size, inflates to its ISIZE and CRC, the header is the SAM header with no references, every record is the SAM record
    of the same run field by field — name, flag, sequence, qualities, the tags in Auxiliary::encode's order (auxiliary.cpp:320-354)
    with FI / TC as int32 — with tid, pos, mate -1, bin, mapq, tlen 0 as the reference leaves them (hts.cpp:26-44), and XB / XM / XC
    are the floats whose `%g` the SAM line prints."""
    import bam_reader
    directory = str(tmp_path)
    if case == "mixed_tokens":
        spec, lengths, n = _mixed_spec(), (40, 8, 26), 6007
    else:
        spec, lengths, n = synthetic.config_spec("C2", "pamld"), (30, 8), 20011
        spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    paths = _files(directory, spec, n, 9, lengths)
    spec["input"] = paths
    outputs = {}
    for extension, host in (("sam", False), ("bam", True), ("bam", False)):
        job = json.loads(json.dumps(spec))
        job["output"] = [os.path.join(directory, "out." + extension)]
        if case == "per_barcode":
            for k, record in enumerate(job["sample"]["codec"].values()):
                if k % 3 == 0:
                    record["output"] = [os.path.join(directory, "barcode_%d.%s" % (k, extension))]
        incoming, _, did = _run(job, 4096, host)
        assert incoming[0] == n
        # the lines of a BAM sink are assembled on the device too (XB / XM / XC as the float's bits) and made records on the host's threads
        assert (did["reads"] == n and did["record_bytes"] > 0) if not host else did["record_bytes"] == 0, did
        outputs[extension] = sorted(name for name in os.listdir(directory) if name.endswith("." + extension))
        if extension == "bam" and host:
            from_host = {name: bam_reader.blocks(open(os.path.join(directory, name), "rb").read())[0] for name in outputs["bam"]}
            for name in outputs["bam"]:
                os.unlink(os.path.join(directory, name))
    # the same bytes in the blocks whichever side assembled the lines
    assert from_host == {name: bam_reader.blocks(open(os.path.join(directory, name), "rb").read())[0] for name in outputs["bam"]}
    assert [name[:-4] for name in outputs["sam"]] == [name[:-4] for name in outputs["bam"]] and (len(outputs["bam"]) > 1) == (case == "per_barcode")
    total = 0
    for sam_name, bam_name in zip(outputs["sam"], outputs["bam"]):
        lines = open(os.path.join(directory, sam_name)).read().split("\n")
        header = "".join(line + "\n" for line in lines if line.startswith("@"))
        body = [line for line in lines if line and not line.startswith("@")]
        text, records, blocks = bam_reader.records(os.path.join(directory, bam_name))
        assert text == header and len(records) == len(body)
        total += len(records)
        for line, record in zip(body, records):
            field = line.split("\t")
            assert record["name"] == field[0] and record["flag"] == int(field[1])
            assert (record["ref"], record["pos"], record["mapq"], record["bin"], record["n_cigar"], record["next_ref"], record["next_pos"], record["tlen"]) == (-1, -1, 0, 0, 0, -1, -1, 0)
            assert field[2:9] == ["*", "0", "0", "*", "*", "0", "0"]
            assert record["sequence"] == ("" if field[9] == "*" else field[9])
            assert record["quality"] == (b"" if field[9] == "*" else bytes(ord(c) - 33 for c in field[10]))
            assert len(record["tags"]) == len(field) - 11
            for (tag, kind, value), stated in zip(record["tags"], field[11:]):
                assert stated[:5] == "%s:%s:" % (tag, kind), (stated, tag, kind)
                if kind == "Z":
                    assert value == stated[5:]
                elif kind == "i":
                    assert value == int(stated[5:])
                else:
                    assert tag in ("XB", "XM", "XC") and stated[5:] == "%g" % value, (stated, value)      # the float the line prints six digits of
    assert total >= n
    if case == "mixed_tokens":
        assert total == 4 * n


def test_an_input_out_of_the_ordinary_part_way_is_handed_to_the_host_reader(tmp_path):
    """A carriage return in record 50 000 of 70 001, batches of 4 096 reads: twelve rounds are assembled on the device, the
    thirteenth finds the record, and the host reader — which tolerates it, as the reference does — continues at the byte the
    twelfth ended at.  The file and the counters are those of a run that went through the host from its start."""
    directory = str(tmp_path)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    n, capacity = 70001, 4096
    paths = _files(directory, spec, n, 6, (30, 8))
    lines = open(paths[0], "rb").read().split(b"\n")
    lines[4 * 50000 + 1] += b"\r"
    lines[4 * 50000 + 3] += b"\r"
    open(paths[0], "wb").write(b"\n".join(lines))
    spec["input"] = paths
    outputs = {}
    for host in (False, True):
        spec["output"] = [os.path.join(directory, "host.sam" if host else "device.sam")]
        incoming, words, did = _run(spec, capacity, host)
        outputs[host] = (incoming, words, _records(spec["output"][0]), did)
    device, host = outputs[False], outputs[True]
    assert device[3]["resumed"] == 1 and device[3]["rounds"] == 50000 // capacity and device[3]["reads"] == 50000 // capacity * capacity
    assert device[0] == host[0] and device[0][0] == n
    assert np.array_equal(device[1], host[1])
    assert device[2] == host[2]


@pytest.mark.parametrize("case", ["file_per_barcode", "gzip_per_segment", "one_gzip_file"])
def test_channels_in_files_of_their_own_and_gzip_output(tmp_path, case):
    """Multiplexer::push (multiplex.h:198-240): every barcode of the multiplexing decoder is a channel with its own URLs.  On the
    device every output file is a sink: pass 1 sums the bytes of every (workgroup, sink), a scan places them, pass 2 sends each
    record to its sink's part of the text; gzip sinks are compressed by the host, a member per sink and round.  The files must
    hold what the host writer's hold — gzip files after inflating (the members are cut elsewhere)."""
    import gzip
    directory = str(tmp_path)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    n, capacity = 60001, 4096
    paths = _files(directory, spec, n, 21, (30, 8))
    spec["input"] = paths
    keys = list(spec["sample"]["codec"])
    names = []

    def outputs(prefix):
        names.clear()
        if case == "one_gzip_file":
            spec["output"] = [os.path.join(directory, prefix + "_all.sam.gz")]
            names.append(spec["output"][0])
            return
        spec["output"] = [os.path.join(directory, prefix + "_rest.sam")]
        names.append(spec["output"][0])
        for k, key in enumerate(keys[:40]):
            if case == "file_per_barcode":
                # SAM for some, one interleaved FASTQ file for others, two FASTQ files for the rest; one barcode silent, one filtered
                if k % 3 == 0:
                    urls = [os.path.join(directory, "%s_%d.sam" % (prefix, k))]
                elif k % 3 == 1:
                    urls = [os.path.join(directory, "%s_%d.fastq" % (prefix, k))]
                else:
                    urls = [os.path.join(directory, "%s_%d_s%d.fastq" % (prefix, k, segment)) for segment in (1, 2)]
            else:
                urls = [os.path.join(directory, "%s_%d_s%d.fastq.gz" % (prefix, k, segment)) for segment in (1, 2)]
            if k == 5:
                urls = ["/dev/null"]
            spec["sample"]["codec"][key]["output"] = urls
            spec["sample"]["codec"][key]["filter outgoing qc fail"] = k == 7
            names.extend(u for u in urls if u != "/dev/null")

    results = {}
    for host in (False, True):
        outputs("host" if host else "device")
        incoming, words, did = _run(spec, capacity, host)
        content = {}
        for name in names:
            data = open(name, "rb").read()
            if name.endswith(".gz"):
                data = gzip.decompress(data)
            content[os.path.basename(name).split("_", 1)[1]] = b"".join(line for line in data.split(b"\n") if not line.startswith(b"@PG"))
        results[host] = (incoming, words, did, content)
    device, host = results[False], results[True]
    assert device[2]["rounds"] == -(-n // capacity) and device[2]["record_bytes"] > 0 and host[2]["rounds"] == 0
    assert device[0] == host[0] and np.array_equal(device[1], host[1])
    assert set(device[3]) == set(host[3]) and len(device[3]) == len(names)
    for name in device[3]:
        assert device[3][name] == host[3][name], name
    assert sum(len(v) for v in device[3].values()) > 100 * n


@pytest.mark.parametrize("host", [False, True])
def test_files_per_barcode_equal_the_independent_python_restatement(oracle, tmp_path, host):
    """The layout of test_channels_in_files_of_their_own_and_gzip_output ("file_per_barcode": a SAM file for some barcodes, one
    interleaved FASTQ file for others, a FASTQ file per output segment for the rest, one barcode silent, one with its own outgoing
    filter, everything else and the unclassified reads in the job's SAM file) against tests/sam_restatement.py over the ORACLE's
    decisions: which file a read's records go to, in input order, and what they say."""
    import sam_restatement
    directory = str(tmp_path)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    n, capacity = 20011, 4096
    paths = _files(directory, spec, n, 23, (30, 8))
    spec["input"] = paths
    keys = list(spec["sample"]["codec"])
    spec["output"] = [os.path.join(directory, "rest.sam")]
    urls_of = {}
    for k, key in enumerate(keys[:40]):
        urls = ["%d.sam" % k] if k % 3 == 0 else ["%d.fastq" % k] if k % 3 == 1 else ["%d_s1.fastq" % k, "%d_s2.fastq" % k]
        urls_of[k] = [] if k == 5 else urls
        spec["sample"]["codec"][key]["output"] = ["/dev/null"] if k == 5 else [os.path.join(directory, u) for u in urls]
        spec["sample"]["codec"][key]["filter outgoing qc fail"] = k == 7
    incoming, _, did = _run(spec, capacity, host)
    assert incoming[0] == n and (did["record_bytes"] == 0) == host

    job = capi.compile_job(spec)
    ordinal_of_index = {job["sample"]["codec"][key]["index"]: k for k, key in enumerate(keys)}
    files = [sam_restatement.read_fastq(path) for path in paths]
    engine = oracle.OracleJob(job)
    base, quality = engine.stage([[(files[i][k][2], files[i][k][3]) for i in range(len(files))] for k in range(n)])
    flags = np.array([1 if files[0][k][1] else 0 for k in range(n)], dtype=np.uint8)
    result, word, _, confidence = engine.classify(base, quality, flags, summary=True)
    expected = {}
    for channel, qcfail, sam, fastq in sam_restatement.by_channel(job, files, result, word, confidence):
        k = ordinal_of_index.get(channel, -1) if channel > 0 else -1
        if k < 0 or k >= 40:
            expected.setdefault("rest.sam", []).extend(sam)
            continue
        if k == 7 and qcfail:
            continue
        urls = urls_of[k]
        if len(urls) == 1:
            expected.setdefault(urls[0], []).extend(sam if urls[0].endswith(".sam") else fastq)
        for segment, url in enumerate(urls if len(urls) == 2 else []):
            expected.setdefault(url, []).append(fastq[segment])
    written = 0
    for name, lines in expected.items():
        data = open(os.path.join(directory, name), "rb").read()
        text = ("\n".join(lines) + "\n").encode()
        if name.endswith(".sam"):
            _assert_sam_equal_but_for_posterior_digits(b"".join(line for line in data.splitlines(True) if not line.startswith(b"@")), text)
        else:
            assert data == text, name
        written += 1
    assert written >= 50 and not os.path.exists(os.path.join(directory, "5.sam"))
    # a barcode's files exist even where no read went to it; nothing but headers in them then
    for k, urls in urls_of.items():
        for url in urls:
            if url not in expected:
                data = open(os.path.join(directory, url), "rb").read()
                assert all(line.startswith(b"@") for line in data.splitlines()) and not url.endswith(".fastq") or data == b""


@pytest.mark.parametrize("case", ["written", "report_only", "mixed_tokens", "filters"])
def test_quality_control_accumulators_on_the_device(tmp_path, case):
    """`enable quality control` (multiplex.h:28-181): the accumulators of the output reads — per channel, output segment, cycle,
    nucleotide and quality, and the reads' mean qualities — kept by a kernel (record_quality_control) must be word for word the
    image the host's record threads leave (tests/test_gpu_quality_control.py holds that one against a numpy restatement of the
    reference's accumulators): with the records written, with every output /dev/null, with corrected barcodes and `~` in the
    template, and with reads the pull loop drops and reads the outgoing filter keeps out of the file (counted all the same,
    multiplex.h:219-221)."""
    directory = str(tmp_path)
    if case == "mixed_tokens":
        spec = _mixed_spec()
        lengths, n, capacity = (40, 8, 26), 20011, 1 << 13
    else:
        spec = synthetic.config_spec("C2", "pamld")
        spec["template"] = {"transform": {"token": ["0::", "1::"]}}
        lengths, n, capacity = (30, 8), 50001, 4096
        if case == "filters":
            spec["filter incoming qc fail"] = True
            spec["filter outgoing qc fail"] = True
    spec["enable quality control"] = True
    paths = _files(directory, spec, n, 11, lengths)
    if case == "filters":
        # reads of two lengths: the cells grow in the run, the shortest and the longest segment differ
        lines = open(paths[0], "rb").read().split(b"\n")
        for i in range(0, n, 7):
            lines[4 * i + 1] = lines[4 * i + 1][:21]
            lines[4 * i + 3] = lines[4 * i + 3][:21]
        open(paths[0], "wb").write(b"\n".join(lines))
    spec["input"] = paths
    outputs = {}
    for host in (False, True):
        spec["output"] = ["/dev/null"] if case == "report_only" else [os.path.join(directory, "host.sam" if host else "device.sam")]
        incoming, words, did = _run(spec, capacity, host)
        outputs[host] = (incoming, words, did, None if case == "report_only" else _records(spec["output"][0]))
    device, host = outputs[False], outputs[True]
    assert device[2]["rounds"] > 1 and host[2]["rounds"] == 0 and device[2]["resumed"] == 0
    assert device[0] == host[0] and np.array_equal(device[1], host[1]) and device[3] == host[3]
    mine, theirs = device[2]["quality_control"], host[2]["quality_control"]
    assert mine.size == theirs.size > 1000 and int(mine[0]) > 1
    assert np.array_equal(mine, theirs)
    assert int(mine.sum()) > n


def _random_template(rng, lengths):
    """a template of 1 to 4 output segments over random tokens: slices of the input segments (open ends, offsets from the end,
    empty ones) and the corrected sample / cellular / molecular barcodes, some reverse complemented"""
    tokens, knit = [], []
    for _ in range(int(rng.integers(1, 5))):
        pieces = []
        for _ in range(int(rng.integers(1, 4))):
            if rng.random() < 0.3:
                source = str(rng.choice(["s", "c", "m"]))
                start = "" if rng.random() < 0.5 else str(int(rng.integers(0, 4)))
                end = "" if rng.random() < 0.5 else str(int(rng.integers(3, 12)))
            else:
                segment = int(rng.integers(0, len(lengths)))
                source = str(segment)
                kind = rng.integers(0, 5)
                if kind == 0:
                    start, end = "", ""
                elif kind == 1:
                    start, end = str(int(rng.integers(0, lengths[segment]))), ""
                elif kind == 2:
                    start, end = str(-int(rng.integers(1, lengths[segment]))), ""
                elif kind == 3:
                    first = int(rng.integers(0, lengths[segment]))
                    start, end = str(first), str(int(rng.integers(first, lengths[segment] + 3)))
                else:
                    start, end = "", str(-int(rng.integers(1, lengths[segment])))
            tokens.append("%s:%s:%s" % (source, start, end))
            pieces.append(("~" if rng.random() < 0.3 else "") + str(len(tokens) - 1))
        knit.append(":".join(pieces))
    return {"transform": {"token": tokens, "knit": knit}}


@pytest.mark.parametrize("seed", range(16))
def test_random_templates_on_the_device_equal_the_host_writers(tmp_path, seed):
    """Sixteen drawn jobs over the five decoder job of _mixed_spec: a random template (1 to 4 output segments, slices with open and
    negative ends, corrected barcodes, `~`), SAM or FASTQ records, the pre-filters, the outgoing filter and the quality control
    accumulators on or off, and half of the time six barcodes in files of their own (SAM, FASTQ, a FASTQ file per output segment,
    plain or gzip).  Records byte for byte, counters and accumulators word for word the host's."""
    rng = np.random.default_rng(1000 + seed)
    directory = str(tmp_path)
    spec = _mixed_spec()
    lengths = (40, 8, 26)
    spec["template"] = _random_template(rng, lengths)
    if rng.random() < 0.4:
        spec["filter incoming qc fail"] = True
    if rng.random() < 0.4:
        spec["filter outgoing qc fail"] = True
    if rng.random() < 0.5:
        spec["enable quality control"] = True
    n, capacity = 9001, 4096
    paths = _files(directory, spec, n, 100 + seed, lengths)
    spec["input"] = paths
    extension = ".fastq" if rng.random() < 0.4 else ".sam"
    own_files = rng.random() < 0.5                       # some barcodes in files of their own: one per output segment, or one, plain or gzip
    n_out = len(spec["template"]["transform"]["knit"])
    choices = [(int(rng.integers(0, 3)), bool(rng.random() < 0.3)) for _ in range(6)]
    outputs = {}
    for host in (False, True):
        prefix = "host" if host else "device"
        spec["output"] = [os.path.join(directory, prefix + "_rest" + extension)]
        names = [spec["output"][0]]
        if own_files:
            for k, key in enumerate(list(spec["sample"]["codec"])[:6]):
                kind, zipped = choices[k]
                tail = (".sam" if kind == 0 else ".fastq") + (".gz" if zipped else "")
                urls = [os.path.join(directory, "%s_%d%s" % (prefix, k, tail))] if kind < 2 else \
                       [os.path.join(directory, "%s_%d_s%d%s" % (prefix, k, o, tail)) for o in range(n_out)]
                spec["sample"]["codec"][key]["output"] = urls
                names.extend(urls)
        incoming, words, did = _run(spec, capacity, host)
        content = {}
        for name in names:
            data = open(name, "rb").read()
            if name.endswith(".gz"):
                import gzip
                data = gzip.decompress(data)
            content[os.path.basename(name).split("_", 1)[1]] = b"\n".join(line for line in data.split(b"\n") if not line.startswith(b"@PG"))
        outputs[host] = (incoming, words, did, content)
    device, host = outputs[False], outputs[True]
    assert device[2]["rounds"] == 3 and device[2]["resumed"] == 0 and host[2]["rounds"] == 0, spec["template"]
    assert device[0] == host[0] and np.array_equal(device[1], host[1]), spec["template"]
    assert set(device[3]) == set(host[3])
    for name in device[3]:
        if device[3][name] != host[3][name]:
            for k, (a, b) in enumerate(zip(device[3][name].split(b"\n"), host[3][name].split(b"\n"))):
                assert a == b, (name, k, a, b, spec["template"])
        assert device[3][name] == host[3][name], (name, spec["template"])
    assert np.array_equal(device[2]["quality_control"], host[2]["quality_control"]), spec["template"]


@pytest.mark.parametrize("oddity", ["carriage returns from a record on", "a blank line between two records", "no last newline", "gzip input, carriage returns"])
def test_what_the_host_reader_tolerates_part_way_gives_the_hosts_files(tmp_path, oddity):
    """Things the reference's reader tolerates (and the device path does not look at: it hands over), met in the middle of a run
    that writes a file per barcode with the quality control accumulators on: the rounds before are the device's, the rest the
    host's, every file and every accumulator what a run through the host from the start leaves."""
    import gzip
    directory = str(tmp_path)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    spec["enable quality control"] = True
    n, capacity = 40001, 4096
    paths = _files(directory, spec, n, 31, (30, 8))
    lines = open(paths[1], "rb").read().split(b"\n")
    if oddity.endswith("carriage returns from a record on") or oddity == "gzip input, carriage returns":
        for k in range(4 * 23000, len(lines) - 1):
            lines[k] += b"\r"
    elif oddity == "a blank line between two records":
        lines.insert(4 * 23000, b"")
    text = b"\n".join(lines)
    if oddity == "no last newline":
        text = text[:-1]
    open(paths[1], "wb").write(text)
    if oddity == "gzip input, carriage returns":
        with gzip.open(paths[1] + ".gz", "wb", compresslevel=1) as packed:
            packed.write(text)
        paths[1] += ".gz"
    spec["input"] = paths
    keys = list(spec["sample"]["codec"])
    results = {}
    for host in (False, True):
        prefix = "host" if host else "device"
        spec["output"] = [os.path.join(directory, prefix + "_rest.sam")]
        names = [spec["output"][0]]
        for k, key in enumerate(keys[:10]):
            spec["sample"]["codec"][key]["output"] = [os.path.join(directory, "%s_%d.fastq" % (prefix, k))]
            names.append(spec["sample"]["codec"][key]["output"][0])
        incoming, words, did = _run(spec, capacity, host)
        results[host] = (incoming, words, did, {os.path.basename(name).split("_", 1)[1]: _records(name) for name in names})
    device, host = results[False], results[True]
    if oddity == "no last newline":
        assert device[2]["resumed"] == 0 and device[2]["rounds"] == -(-n // capacity)          # the device adds the newline itself
    else:
        assert device[2]["resumed"] == 1 and device[2]["rounds"] == 23000 // capacity
    assert device[0] == host[0] and device[0][0] == n
    assert np.array_equal(device[1], host[1])
    assert device[3] == host[3]
    assert np.array_equal(device[2]["quality_control"], host[2]["quality_control"])


@pytest.mark.parametrize("damage", ["names out of sync", "a file cut inside a record"])
def test_damage_part_way_raises_the_host_readers_error(tmp_path, damage):
    """names out of sync in record 30 000, a file that ends inside record 30 000: the device path hands over at the round that
    holds it and the host reader raises what it raises when it reads the files from the start"""
    directory = str(tmp_path)
    spec = synthetic.config_spec("C2", "pamld")
    spec["template"] = {"transform": {"token": ["0::", "1::"]}}
    n, capacity = 50001, 4096
    paths = _files(directory, spec, n, 32, (30, 8))
    data = open(paths[1], "rb").read()
    lines = data.split(b"\n")
    if damage == "names out of sync":
        lines[4 * 30000] = lines[4 * 30000].replace(b"@M:1:", b"@X:1:", 1)
        open(paths[1], "wb").write(b"\n".join(lines))
    else:
        open(paths[1], "wb").write(b"\n".join(lines[:4 * 30000 + 2]) + b"\nFFF")
    spec["input"] = paths
    raised = {}
    for host in (False, True):
        spec["output"] = [os.path.join(directory, "host.sam" if host else "device.sam")]
        if host:
            os.environ["PHQ_HOST_RECORDS"] = "1"
        try:
            before = capi.device_path_counters()
            s = DecoderSet(json.dumps(spec), device=0)
            with pytest.raises(PheniqsError) as caught:
                s.transcode_job(capacity)
            raised[host] = (caught.value.code, str(caught.value), capi.device_path_counters()["resumed"] - before["resumed"])
            s.close()
        finally:
            os.environ.pop("PHQ_HOST_RECORDS", None)
    assert raised[False][0] == raised[True][0] and raised[False][1] == raised[True][1]
    assert raised[False][2] == 1 and raised[True][2] == 0
