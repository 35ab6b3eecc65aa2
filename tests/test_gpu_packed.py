"""phq_pack_planes / phq_classify_host_packed: one byte per nucleotide across the link (csrc/unpack.hip).  The packed call must
return the bytes phq_classify_host returns — records, status words, counters — whatever the batch holds: N and the other
ambiguity codes, qualities of 63 and more, reads shorter than the barcode (zero padding), several decoders, several chunks."""
import numpy as np
import pytest

from pheniqs_amd import DecoderSet, PheniqsError, capi, synthetic

pytestmark = pytest.mark.gpu


def _both(spec, code, quality, flag, offsets, capacity=None):
    n = code.shape[0]
    out = []
    for packed in (False, True):
        s = DecoderSet(spec, device=0)
        s.stats_reset()
        base = s.planes_from_rows(code, offsets, capacity)
        qual = s.planes_from_rows(quality, offsets, capacity)
        if packed:
            planes, escapes = s.pack_planes(n, base, qual, capacity)
            result, word = s.classify_host_packed(n, planes, escapes, flag, capacity)
            out.append((result, word, s.stats_export(), escapes))
        else:
            result, word = s.classify_host(n, base, qual, flag, capacity)
            out.append((result, word, s.stats_export(), None))
        s.close()
    return out


@pytest.mark.parametrize("config,n", [("C2", 300001), ("C4", 120000), ("C5", 90000), ("C3", 100000)])
def test_packed_planes_give_the_same_bytes(config, n):
    spec = synthetic.config_spec(config, "pamld")
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, 17)
    rng = np.random.default_rng(3)
    # what does not pack: other ambiguity codes, qualities of 63 and more, reads cut short (zero padding to the end)
    odd = rng.random(code.shape) < 0.002
    code[odd] = rng.integers(0, 16, size=int(odd.sum()), dtype=np.uint8)
    high = rng.random(code.shape) < 0.002
    quality[high] = rng.integers(63, 94, size=int(high.sum()), dtype=np.uint8)
    short = rng.random(n) < 0.01
    code[short, -3:] = 0
    quality[short, -3:] = 0
    plain, packed = _both(spec, code, quality, flag, offsets)
    assert packed[3].size > 0
    assert np.array_equal(plain[0].view(np.uint8), packed[0].view(np.uint8))
    assert np.array_equal(plain[1], packed[1])
    assert np.array_equal(plain[2], packed[2])


def test_packed_planes_of_a_larger_capacity_and_several_chunks():
    """planes laid out for more reads than are classified; 5 M reads are three chunks of 2 Mi"""
    spec = synthetic.config_spec("C2", "pamld")
    job = capi.compile_job(spec)
    pool = 1 << 18
    code, quality, flag, offsets = synthetic.generate_rows(job, pool, 5)
    n = 5_000_000
    reps = -(-n // pool)
    code, quality, flag = np.tile(code, (reps, 1))[:n], np.tile(quality, (reps, 1))[:n], np.tile(flag, reps)[:n].copy()
    plain, packed = _both(spec, code, quality, flag, offsets, capacity=n + 12345)
    assert np.array_equal(plain[0].view(np.uint8), packed[0].view(np.uint8))
    assert np.array_equal(plain[1], packed[1])
    assert np.array_equal(plain[2], packed[2])


def test_the_escape_list_is_checked():
    spec = synthetic.config_spec("C2", "pamld")
    job = capi.compile_job(spec)
    code, quality, flag, offsets = synthetic.generate_rows(job, 4096, 1)
    s = DecoderSet(spec, device=0)
    base, qual = s.planes_from_rows(code, offsets), s.planes_from_rows(quality, offsets)
    planes, escapes = s.pack_planes(4096, base, qual)
    assert escapes.size > 1 and np.all(np.diff((escapes >> np.uint64(32)).astype(np.int64)) >= 0)
    with pytest.raises(PheniqsError) as caught:
        s.classify_host_packed(4096, planes, escapes[::-1].copy(), flag)
    assert "read order" in str(caught.value)
    # without the status word
    result, word = s.classify_host_packed(4096, planes, escapes, flag, with_word=False)
    expected, _ = s.classify_host(4096, base, qual, flag)
    assert word is None and np.array_equal(result.view(np.uint8), expected.view(np.uint8))
    s.close()
