"""A second, independent restatement — in Python, from the reference's text — of what the reference WRITES for a read: the SAM
records of its output segments.  Test infrastructure: it formats, it decides nothing; which barcode a decoder named, the QC
flag and the per type confidences come from the oracle (oracle/pheniqs_oracle.c) run on the same reads.

    transcode.h:207-228      the pull loop: incoming filters, name and QC flag onto every output segment, classify, template, flush
    naive.h:41-46            NaiveMolecularDecoder: the raw molecular barcode only
    pamld.cpp:133-180        Paml{Sample,Cellular,Molecular}Decoder::classify: raw + corrected barcode, RG
    mdd.cpp:96-138           Md{Sample,Cellular,Molecular}Decoder::classify: the same
    transform.h:65-80        Token::absolute_start / absolute_end
    transform.h:142-169      Rule::apply (a decoder's observation) — restated here as the template's apply, on observations
    transform.h:190-226      TemplateRule::apply: tokens of input segments and of the corrected barcodes (s, c, m), `~`
    sequence.h:382-398       ObservedSequence::append_corrected, its `original.code[length + i]` probe kept
    read.h:187-237           Read::flush: XB / XM / XC = float(1 - confidence) where 0 < confidence < 1, the barcode tags on
                             every segment, the leader's QC flag, XB / XM / XC / RG copied to the other segments
    read.h:349-377           flags: unmapped + mate unmapped, paired when there are two segments or more, READ1 on the first,
                             READ2 on the last; FI / TC
    auxiliary.cpp:320-354    Auxiliary::encode: FI TC (TC > 2 only) RG BC QT XB RX QX OX BZ XM CB CR CY XC, in that order
    hts.h:169-239            the record: name, flag, no reference, no position, no cigar, SEQ, QUAL; htslib prints a float tag as %g

    fastq.h:144-194          a FASTQ record of an output segment and its Illumina comment (`fastq_records()`)

`records()` is held against the reference's own golden SAM file (tests/test_sam_restatement.py, no GPU) before anything of the
library's is held against it.
"""
import numpy as np

IUPAC = "=ACMGRSVTWYHKDBN"           # BAM code -> letter (iupac.h:103-121)
ASCII_TO_BAM = np.full(256, 15, dtype=np.uint8)        # iupac.h:153-171: anything unknown reads as N
for _code, _letter in enumerate(IUPAC):
    ASCII_TO_BAM[ord(_letter)] = _code
    ASCII_TO_BAM[ord(_letter.lower())] = _code
# the complement of a set of bases is the set of their complements: A (1) <-> T (8), C (2) <-> G (4)   (iupac.h: BamToReverseComplementBam)
COMPLEMENT = np.array([((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3) for c in range(16)], dtype=np.uint8)

HEADS = {"s": -1, "c": -2, "m": -3}
TAG_ORDER = ("FI", "TC", "RG", "BC", "QT", "XB", "RX", "QX", "OX", "BZ", "XM", "CB", "CR", "CY", "XC")


def read_fastq(path):
    """[(name, qc fail, BAM codes, Phred values)] of a FASTQ file (fastq.h:55-78; the Illumina comment's filter flag, fastq.h:253-285)"""
    lines = open(path, "rb").read().split(b"\n")
    records = []
    for k in range(0, len(lines) - 3, 4):
        name, _, comment = lines[k][1:].partition(b" ")
        fields = comment.split(b":")
        qcfail = len(fields) > 1 and fields[1][:1] == b"Y"
        code = ASCII_TO_BAM[np.frombuffer(lines[k + 1], dtype=np.uint8)]
        quality = np.frombuffer(lines[k + 3], dtype=np.uint8) - 33
        records.append((name.decode(), qcfail, code, quality.astype(np.uint8)))
    return records


def transforms_of(transform):
    """[(input segment or -1 / -2 / -3, start, end, end terminated, output segment, reverse complement)] of token + knit
    (transform.cpp:118-174,252-331); no knit = every token an output segment of its own"""
    tokens = []
    for pattern in transform["token"]:
        head, start, end = pattern.split(":")
        tokens.append((HEADS[head] if head in HEADS else int(head), int(start) if start else 0, int(end) if end else 0, end != ""))
    out = []
    for output_segment, pattern in enumerate(transform.get("knit") or [str(k) for k in range(len(tokens))]):
        for piece in pattern.split(":"):
            reverse = piece.startswith("~")
            out.append(tokens[int(piece[1:] if reverse else piece)] + (output_segment, reverse))
    return out


def _slice(token, code, quality):
    """a token's bases of one sequence: transform.h:65-80 and the two branches of the apply functions"""
    _, start, end, end_terminated, _, reverse = token
    length = len(code)
    if end_terminated:
        stop = max(length + end, 0) if end < 0 else min(end, length)
    else:
        stop = length
    first = max(length + start, 0) if start < 0 else (0 if start > length else start)
    if stop - first <= 0:
        return code[:0], quality[:0]
    if reverse:
        return COMPLEMENT[code[first:stop][::-1]], quality[first:stop][::-1]
    return code[first:stop], quality[first:stop]


def _apply(transforms, source, corrected=None):
    """the output segments of a rule as [(codes, qualities)]"""
    cardinality = 1 + max(t[4] for t in transforms)
    codes = [[] for _ in range(cardinality)]
    qualities = [[] for _ in range(cardinality)]
    for token in transforms:
        if token[0] < 0:
            code, quality = corrected[{-1: "sample", -2: "cellular", -3: "molecular"}[token[0]]]
        else:
            code, quality = source[token[0]]
        c, q = _slice(token, code, quality)
        codes[token[4]].append(c)
        qualities[token[4]].append(q)
    empty = np.zeros(0, dtype=np.uint8)
    return [(np.concatenate(c) if c else empty, np.concatenate(q) if q else empty) for c, q in zip(codes, qualities)]


def _letters(code):
    return "".join(IUPAC[c] for c in code)


def _phred(quality):
    return "".join(chr(33 + q) for q in quality)


def decoders_of(job):
    """(kind, decoder) in the order TranscodingDecoder::classify calls them (transcode.h:51-60)"""
    out = []
    for kind in ("sample", "molecular", "cellular"):
        node = job.get(kind)
        if node is None:
            continue
        for decoder in ([node] if isinstance(node, dict) else node):
            out.append((kind, decoder))
    return out


def _reads(job, files, result, word, confidence, filtered=None):
    """(name, QC fail, tags, output segments) of the reads of `files` (one FASTQ file per input segment, as read_fastq returns
    them) that pass the incoming filters; result[k][r] = (index, ...) of the r-th PAMLD / MDD decoder for the k-th read that
    PASSED them, word[k] its status word (bit 31: QC fail), confidence[k] = (sample, molecular, cellular) as the wrappers
    leave them.  `filtered`, a list, receives the ordinals of the reads the incoming filters dropped."""
    decoders = decoders_of(job)
    template = transforms_of(job["template"]["transform"])
    minimum = job.get("min input length") or []
    k = 0
    for ordinal, segments in enumerate(zip(*files)):
        name, qcfail_in = segments[0][0], segments[0][1]
        # transcode.h:193-211: the incoming filters (the minimum length is checked from segment 1 on)
        if (job.get("filter incoming qc fail") and qcfail_in) or any(len(segments[i][2]) < minimum[i] for i in range(1, min(len(minimum), len(segments)))):
            if filtered is not None:
                filtered.append(ordinal)
            continue
        source = [(s[2], s[3]) for s in segments]
        raw = {kind: ([], []) for kind in ("sample", "molecular", "cellular")}
        corrected = {kind: ([], []) for kind in ("sample", "molecular", "cellular")}
        group = None
        r = 0
        for kind, decoder in decoders:
            algorithm = decoder.get("algorithm", "passthrough")
            if algorithm == "passthrough":
                continue
            observation = _apply(transforms_of(decoder["transform"]), source)
            for code, quality in observation:                       # append_to_raw_*_barcode
                raw[kind][0].extend(code)
                raw[kind][1].extend(quality)
            if algorithm == "naive":
                continue
            index = int(result[k][r]["index"])
            r += 1
            entry = decoder["undetermined"]
            if index > 0:
                entry = next(e for e in decoder["codec"].values() if e["index"] == index)
            if kind == "sample":
                group = entry["ID"]
            # append_to_corrected_*_barcode_sequence: segment by segment, the probe of sequence.h:389 at the DESTINATION's length
            to_code, to_quality = corrected[kind]
            for (code, quality), word_text in zip(observation, entry["barcode"]):
                if len(code) == 0:
                    continue
                fixed = ASCII_TO_BAM[np.frombuffer(word_text.encode(), dtype=np.uint8)]
                length = len(to_code)
                for i in range(len(code)):
                    probe = code[length + i] if length + i < len(code) else 0          # past the observation: its terminator (S9)
                    to_code.append(int(fixed[i]))
                    to_quality.append(int(quality[i]) if (probe == fixed[i] or fixed[i] == 0) else int(decoder["corrected quality"]))
        as_arrays = {kind: (np.array(c, dtype=np.uint8), np.array(q, dtype=np.uint8)) for kind, (c, q) in corrected.items()}
        output = _apply(template, source, as_arrays)
        qcfail = bool(int(word[k]) >> 31)
        tags = {}
        if group is not None:
            tags["RG"] = "Z:" + group
        for kind, tag in (("sample", "XB"), ("molecular", "XM"), ("cellular", "XC")):
            c = float(confidence[k][{"sample": 0, "molecular": 1, "cellular": 2}[kind]])
            if 0 < c < 1:
                value = float(np.float32(1.0 - c))
                if value > 0:
                    tags[tag] = "f:%g" % value
        if raw["sample"][0]:
            tags["BC"], tags["QT"] = "Z:" + _letters(raw["sample"][0]), "Z:" + _phred(raw["sample"][1])
        if raw["cellular"][0]:
            tags["CR"], tags["CY"] = "Z:" + _letters(raw["cellular"][0]), "Z:" + _phred(raw["cellular"][1])
        if corrected["cellular"][0]:
            tags["CB"] = "Z:" + _letters(corrected["cellular"][0])
        if raw["molecular"][0]:
            tags["OX"], tags["BZ"] = "Z:" + _letters(raw["molecular"][0]), "Z:" + _phred(raw["molecular"][1])
        if corrected["molecular"][0]:
            tags["RX"], tags["QX"] = "Z:" + _letters(corrected["molecular"][0]), "Z:" + _phred(corrected["molecular"][1])
        yield name, qcfail, tags, output
        k += 1


def _sam_lines(name, qcfail, tags, output):
    """hts.h:169-239, read.h:349-377, auxiliary.cpp:320-354: a line (no line break) per output segment"""
    cardinality = len(output)
    lines = []
    for segment, (code, quality) in enumerate(output):
        flag = 4 | 8 | (1 if cardinality > 1 else 0) | (64 if segment == 0 else 0) | (128 if cardinality > 1 and segment == cardinality - 1 else 0) | (512 if qcfail else 0)
        own = dict(tags)
        if cardinality > 2:
            own["FI"], own["TC"] = "i:%d" % (segment + 1), "i:%d" % cardinality
        fields = [name, str(flag), "*", "0", "0", "*", "*", "0", "0", _letters(code) if len(code) else "*", _phred(quality) if len(code) else "*"]
        fields += ["%s:%s" % (tag, own[tag]) for tag in TAG_ORDER if tag in own]
        lines.append("\t".join(fields))
    return lines


def _fastq_lines(illumina, name, qcfail, tags, output):
    """FastqRecord::decode(segment) + encode, fastq.h:79-88,144-171; on the platforms whose comment has that form (fastq.h:176-194)
    the comment is `segment:Y|N:0:raw sample barcode`: a record (four lines, no last line break) per output segment"""
    lines = []
    for segment, (code, quality) in enumerate(output):
        comment = "%d:%s:0:%s" % (segment + 1, "Y" if qcfail else "N", tags["BC"][2:] if "BC" in tags else "") if illumina else ""
        lines.append("@%s %s\n%s\n+\n%s" % (name, comment, _letters(code), _phred(quality)))
    return lines


def records(job, files, result, word, confidence, filtered=None):
    """The SAM records (text, no header) of those reads"""
    lines = []
    for read in _reads(job, files, result, word, confidence, filtered):
        lines.extend(_sam_lines(*read))
    return ("\n".join(lines) + "\n").encode() if lines else b""


def fastq_records(job, files, result, word, confidence, filtered=None):
    """The same reads as FASTQ records, every output segment of a read one after the other (an interleaved file)"""
    illumina = job.get("platform", "ILLUMINA") in ("ILLUMINA", "ELEMENT")
    lines = []
    for read in _reads(job, files, result, word, confidence, filtered):
        lines.extend(_fastq_lines(illumina, *read))
    return ("\n".join(lines) + "\n").encode() if lines else b""


def by_channel(job, files, result, word, confidence):
    """[(channel, QC fail, SAM line per output segment, FASTQ record per output segment)] of the reads that pass the incoming
    filters, in input order: Multiplexer::push (multiplex.h:198-240) sends a read to the channel of the multiplexing decoder's
    barcode — the channel index the status word carries (classifier.h:83-85) —, its URLs and its own outgoing filter"""
    illumina = job.get("platform", "ILLUMINA") in ("ILLUMINA", "ELEMENT")
    out = []
    for k, read in enumerate(_reads(job, files, result, word, confidence)):
        out.append((int(word[k]) & 0x7fffffff, read[1], _sam_lines(*read), _fastq_lines(illumina, *read)))
    return out
