"""the per type accumulators of the output Read (read.h:152-165) that Read::flush turns into XB / XM /
XC, on the boundary as phq_summarize and inside the transcoder, against the oracle's own (orc_classify_summary).

Chains with several decoders of one type: products of confidences, the reset of the cellular distance by an
unclassified cellular decoder (pamld.cpp:149-160), the molecular setter that adds instead (read.h:319-321), and the
MDD wrappers that both reset the CELLULAR distance (mdd.cpp:118,136).
"""
import os

import numpy as np
import pytest

from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic


def chain_job(algorithm, noise):
    """sample + two molecular + three cellular decoders of the same algorithm, noise reads frequent enough that every
    combination of classified / unclassified decoders shows up"""
    rng = np.random.default_rng(77)
    word_sets = [synthetic.barcode_set(rng, count, length) for count, length in ((24, 8), (12, 6), (8, 7), (24, 8), (16, 6), (30, 9))]
    extra = {"noise": noise} if algorithm == "pamld" else {"distance tolerance": [1]}
    tokens = []
    offset = 0
    for words in word_sets:
        tokens.append("0:%d:%d" % (offset, offset + len(words[0])))
        offset += len(words[0])
    decoders = [synthetic.decoder_spec(algorithm, [words], [token], **extra) for words, token in zip(word_sets, tokens)]
    return capi.compile_job({"input segment cardinality": 1, "sample": decoders[0], "molecular": decoders[1:3], "cellular": decoders[3:]})


CASES = [("C4", None, None), ("C5", None, None), ("chain", "pamld", 0.3), ("chain", "mdd", 0.3)]


def build(name, algorithm, noise):
    if name != "chain":
        return synthetic.config_job(name)[0]
    job = chain_job(algorithm, noise)
    return job


def rows_for(job, n, noise):
    if noise is not None:                      # the generator draws noise reads by the decoder's `noise`; MDD jobs carry none
        for decoder in [job.get("sample")] + list(job.get("molecular", [])) + list(job.get("cellular", [])):
            if decoder is not None:
                decoder.setdefault("noise", noise)
    return synthetic.generate_rows(job, n, 4242)


@pytest.mark.parametrize("name,algorithm,noise", CASES)
def test_summarize_host_loop_equals_the_oracle(oracle, name, algorithm, noise):
    """phq_summarize over the ORACLE's result records (no GPU): the host loop alone."""
    job = build(name, algorithm, noise)
    n = 6000
    code, quality, flag, offsets = rows_for(job, n, noise)
    engine = oracle.OracleJob(job)
    expected, word, distance, confidence = engine.classify(code, quality, flag, summary=True)
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    planes = np.ascontiguousarray(expected.T)                   # (n_result, n)
    got_distance, got_confidence = s.summarize(planes)
    assert np.array_equal(got_distance, distance)
    assert np.array_equal(got_confidence, confidence)           # the same records, the same products: bit for bit
    if name == "chain":
        # the cases the quirks are about do occur: an unclassified cellular decoder after a classified one with mismatches
        assert (distance[:, 2] == 0).any() and (distance[:, 2] > 0).any()
        if algorithm == "pamld":
            assert (confidence[:, 2] == 0).any() and ((confidence[:, 2] > 0) & (confidence[:, 2] < 1)).any()
            assert (confidence[:, 1] == 0).any()
        else:
            assert (confidence == 1).all()
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,algorithm,noise", CASES)
def test_summarize_gpu_records_equal_the_oracle(oracle, name, algorithm, noise):
    """The kernels' records through phq_summarize against orc_classify_summary: distances exactly, confidences to the
    posterior tolerance times the number of factors."""
    import torch
    job = build(name, algorithm, noise)
    n = 20000
    code, quality, flag, offsets = rows_for(job, n, noise)
    engine = oracle.OracleJob(job)
    expected, word, distance, confidence = engine.classify(code, quality, flag, summary=True)
    for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
        s = DecoderSet(job, device=0)
        s.set_mode(mode)
        s.stats_reset()
        base = torch.from_numpy(s.planes_from_rows(code, offsets)).cuda()
        qual = torch.from_numpy(s.planes_from_rows(quality, offsets)).cuda()
        flags = torch.from_numpy(flag).cuda()
        result = torch.zeros(s.n_result * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        status = torch.zeros(n, dtype=torch.int32, device="cuda")
        s.classify_device(n, 0, n, base, qual, flags, result, status, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        planes = result.cpu().numpy().view(RESULT_DTYPE).reshape(s.n_result, n)
        got_distance, got_confidence = s.summarize(planes)
        assert np.array_equal(got_distance, distance), (name, algorithm, mode)
        # a type's confidence is the product of up to four decoders' posteriors: each within 1e-12 (literal) / 1e-8 (fast)
        assert np.abs(got_confidence - confidence).max() < (4e-12 if mode == capi.MODE_LITERAL else 4e-8), (name, algorithm, mode)
        assert np.array_equal(got_confidence == 0, confidence == 0) and np.array_equal(got_confidence == 1, confidence == 1)
        s.close()


def _format_g(value):
    """%g of a float, the way htslib prints a SAM 'f' tag"""
    return "%g" % np.float32(value)


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["pamld", "mdd"])
def test_chain_end_to_end_sam_tags_follow_the_oracle(oracle, tmp_path, algorithm):
    """FASTQ text -> phq_transcode_fastq_to_sam -> SAM text for a job with two molecular and three cellular decoders:
    XB / XM / XC (Read::flush, read.h:187-199: 1 - the type's confidence product, as a float), the raw barcodes BC /
    OX / CR and the corrected ones RX / CB of every record are what the oracle's records and per type accumulators
    say.  Under MDD no confidence is ever set, so no X? tag may appear (mdd.cpp:96-138)."""
    import jobspec
    noise = 0.3
    job = chain_job(algorithm, noise)
    n = 30011
    code, quality, flag, offsets = rows_for(job, n, noise)
    engine = oracle.OracleJob(job)
    expected, word, distance, confidence = engine.classify(code, quality, flag, summary=True)

    letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    text = letters[code]
    path = os.path.join(tmp_path, "reads.fastq")
    with open(path, "wb") as out:
        for i in range(n):
            out.write(b"@r%d 1:%s:0:1\n%s\n+\n%s\n" % (i, b"Y" if flag[i] else b"N", text[i].tobytes(), (quality[i] + 33).astype(np.uint8).tobytes()))
    job["template"] = {"transform": {"token": ["0::"]}}
    job = capi.compile_job(job) if "segment" not in job.get("template", {}) else job
    s = DecoderSet(job, device=0)
    output = os.path.join(tmp_path, "out.sam")
    incoming = s.transcode_fastq_to_sam([path], output, 4096)
    assert incoming[0] == n

    decoders = [d for _, d in jobspec.iter_decoders(job)]
    by_index = []
    for d in decoders:
        by_index.append({int(b["index"]): "".join(b["barcode"]) for b in d["codec"].values()})
    kinds = ["sample"] + ["molecular"] * len(job["molecular"]) + ["cellular"] * len(job["cellular"])
    records = [line.rstrip("\n").split("\t") for line in open(output) if not line.startswith("@")]
    assert len(records) == n
    seen = {"XB": 0, "XM": 0, "XC": 0}
    for i, field in enumerate(records):
        tags = {t[:2]: t[5:] for t in field[11:]}
        assert field[0] == "r%d" % i
        raw = {"sample": "", "molecular": "", "cellular": ""}
        corrected = {"sample": "", "molecular": "", "cellular": ""}
        for k, d in enumerate(decoders):
            L = int(d["nucleotide cardinality"])
            observed = text[i, offsets[k]:offsets[k] + L].tobytes().decode()
            raw[kinds[k]] += observed
            index = int(expected[i, k]["index"])
            corrected[kinds[k]] += by_index[k][index] if index > 0 else "=" * L
        assert tags["BC"] == raw["sample"] and tags["OX"] == raw["molecular"] and tags["CR"] == raw["cellular"]
        assert tags["RX"] == corrected["molecular"] and tags["CB"] == corrected["cellular"]
        for tag, kind in (("XB", 0), ("XM", 1), ("XC", 2)):
            c = confidence[i, kind]
            if 0 < c < 1:
                assert tag in tags, (i, tag, c)
                # the GPU's confidence is within 1e-8 of the oracle's: the printed float may differ in its last digit
                assert abs(float(tags[tag]) - float(_format_g(1.0 - c))) <= 1.1e-5 * abs(1.0 - c) + 1e-12, (i, tag, tags[tag], 1.0 - c)      # one unit of %g's sixth digit
                seen[tag] += 1
            else:
                assert tag not in tags, (i, tag, c)
    if algorithm == "pamld":
        assert min(seen.values()) > 1000
    else:
        assert sum(seen.values()) == 0
    s.close()
