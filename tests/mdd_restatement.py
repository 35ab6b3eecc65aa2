"""A second, independent restatement of the reference's minimum distance decoder, in numpy — test infrastructure.

Written from the reference text alone (mdd.cpp:37-86, sequence.h:90-98,321-332, transform.h:65-80,142-169,
iupac.h:107-124), NOT from oracle/pheniqs_oracle.c, and shaped differently on purpose: whole batches as arrays, the
scan as one (reads x barcodes x positions) comparison.  The reference holds no golden vector for MDD, multi segment
barcodes, `~` tokens or quality masking (parity unpinned), so the C oracle's branches for them are
checked by reading; two restatements by different routes that agree on 10^5 random reads a case are the next best
thing (tests/test_mdd_restatement.py)."""
import numpy as np

ASCII_TO_BAM = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
# iupac.h:107-124 BamToReverseComplementBam: the four bits of a code in reverse order
REVERSE_COMPLEMENT = np.array([int("{:04b}".format(c)[::-1], 2) for c in range(16)], dtype=np.uint8)


def parse_rule(transform):
    """token "segment:start:end" (either bound may be missing), knit "[~]token[:[~]token...]" per output segment; no
    knit = token k is output segment k (transform.cpp:118-174,252-331)."""
    tokens = []
    for text in transform["token"]:
        segment, start, end = text.split(":")
        tokens.append((int(segment), int(start) if start else 0, int(end) if end else None))
    knit = transform.get("knit") or [str(k) for k in range(len(tokens))]
    rule = []
    for pattern in knit:
        pieces = []
        for piece in pattern.split(":"):
            flipped = piece.startswith("~")
            pieces.append(tokens[int(piece.lstrip("~"))] + (flipped,))
        rule.append(pieces)
    return rule


def bounds(start, end, length):
    """Token::absolute_start / absolute_end (transform.h:65-80): negative bounds count from the end and stop at 0; a
    start past the end of the segment is 0 (!), an end past it is the length."""
    if start < 0:
        first = max(0, length + start)
    else:
        first = 0 if start > length else start
    if end is None:
        last = length
    elif end < 0:
        last = max(0, length + end)
    else:
        last = min(end, length)
    return first, last


def observe(rule, read):
    """Rule::apply (transform.h:142-169): read = [(code, quality)] per input segment -> [(code, quality)] per output segment."""
    out = []
    for pieces in rule:
        codes, qualities = [], []
        for segment, start, end, flipped in pieces:
            code, quality = read[segment]
            first, last = bounds(start, end, len(code))
            if last - first > 0:
                c, q = code[first:last], quality[first:last]
                if flipped:
                    c, q = REVERSE_COMPLEMENT[c[::-1]], q[::-1]
                codes.append(c)
                qualities.append(q)
        out.append((np.concatenate(codes) if codes else np.zeros(0, np.uint8), np.concatenate(qualities) if qualities else np.zeros(0, np.uint8)))
    return out


def classify(barcodes, segment_lengths, tolerance, masking_threshold, observations):
    """MdDecoder::classify (mdd.cpp:37-86) over a batch.

    barcodes: uint8 [n_barcode, sum(segment_lengths)] BAM codes in index order; observations: one list of (code, quality)
    per read as `observe` gives them.  Returns (decoded [n] with 0 = unclassified else 1 + ordinal, distance [n])."""
    n, n_barcode, width = len(observations), barcodes.shape[0], int(sum(segment_lengths))
    starts = np.concatenate([[0], np.cumsum(segment_lengths)[:-1]]).astype(int)
    code = np.zeros((n, width), dtype=np.uint8)
    quality = np.zeros((n, width), dtype=np.uint8)
    present = np.zeros((n, width), dtype=bool)             # positions the observation really has
    for r, segments in enumerate(observations):
        for k, (c, q) in enumerate(segments):
            take = min(len(c), int(segment_lengths[k]))
            code[r, starts[k]:starts[k] + take] = c[:take]
            quality[r, starts[k]:starts[k] + take] = q[:take]
            present[r, starts[k]:starts[k] + take] = True
    # the map's keys are whole barcodes: only an observation of full length can hit one (mdd.cpp:29-31,43-46)
    whole = present.all(axis=1)
    same = (code[:, None, :] == barcodes[None, :, :]).all(axis=2) & whole[:, None]
    exact = np.where(same.any(axis=1), same.argmax(axis=1) + 1, 0)
    # distance_from / masked_distance_from loop over the OBSERVED length (sequence.h:92,323)
    miss = code[:, None, :] != barcodes[None, :, :]
    if masking_threshold > 0:
        miss = miss | (quality < masking_threshold)[:, None, :]
    miss = miss & present[:, None, :]
    passing = np.ones((n, n_barcode), dtype=bool)
    total = np.zeros((n, n_barcode), dtype=np.int64)
    for k, length in enumerate(segment_lengths):
        error = miss[:, :, starts[k]:starts[k] + int(length)].sum(axis=2)
        passing &= error <= tolerance[k]
        total += error
    first = np.where(passing.any(axis=1), passing.argmax(axis=1) + 1, 0)      # the FIRST barcode that passes, not the nearest (mdd.cpp:75-79)
    decoded = np.where(exact > 0, exact, first)
    distance = np.where(exact > 0, 0, np.where(first > 0, total[np.arange(n), np.maximum(first, 1) - 1], 0))
    return decoded.astype(np.int32), distance.astype(np.int64)
