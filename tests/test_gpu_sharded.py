"""The N > 1 path with the product kernels: reads sharded over ranks, no data path collective, the
counter buffer merged by one SUM all-reduce (transcode.cpp:1816-1818 -> selector.cpp:68-77).

* two ranks on one GPU (started by conftest.py before this process touched the GPU): every rank runs
  phq_classify_device on its shard_range and all-reduces the DEVICE counter tensor; the merged words must equal the
  single process buffer word for word and the concatenated result planes the unsharded ones, byte for byte;
* phq_stats_allreduce, the C++ host's form of that merge, on a one rank RCCL communicator made from the RCCL symbols
  torch loads.
"""
import ctypes
import os

import numpy as np
import pytest

from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,algorithm", [("C2", "pamld"), ("C5", "pamld"), ("C2", "mdd")])
def test_two_ranks_on_one_gpu_equal_the_unsharded_run(sharded_run, config, algorithm):
    import torch
    directory = sharded_run["directory"]
    assert directory and os.path.exists(os.path.join(directory, "done")), "the sharded ranks did not finish:\n" + sharded_run["log"]
    n, world = sharded_run["reads"], sharded_run["world"]
    ranks = [np.load(os.path.join(directory, "%s_%s_rank%d.npz" % (config, algorithm, rank))) for rank in range(world)]

    job, recipe = synthetic.config_job(config, algorithm)
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    s = DecoderSet(job, device=0)
    s.stats_reset()
    base = torch.from_numpy(s.planes_from_rows(code, offsets)).cuda()
    qual = torch.from_numpy(s.planes_from_rows(quality, offsets)).cuda()
    flags = torch.from_numpy(flag).cuda()
    result = torch.zeros(s.n_result * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    word = torch.zeros(n, dtype=torch.int32, device="cuda")
    s.classify_device(n, 0, n, base, qual, flags, result, word, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    single = s.stats_export()
    planes = result.cpu().numpy().view(RESULT_DTYPE).reshape(s.n_result, n)
    words = word.cpu().numpy()

    # the shards tile the batch, in order (output order is reproducible by concatenation)
    assert int(ranks[0]["first"]) == 0 and int(ranks[-1]["last"]) == n
    for a, b in zip(ranks[:-1], ranks[1:]):
        assert int(a["last"]) == int(b["first"])
    # every rank holds the same merged buffer, it is the sum of the ranks' own, and it is the unsharded buffer
    for rank in ranks:
        assert np.array_equal(rank["merged"], ranks[0]["merged"])
    assert np.array_equal(sum(rank["own"].astype(np.uint64) for rank in ranks), ranks[0]["merged"])
    assert np.array_equal(ranks[0]["merged"], single)
    # result planes and status words, byte for byte
    for rank in ranks:
        first, last = int(rank["first"]), int(rank["last"])
        shard = rank["result"].view(RESULT_DTYPE).reshape(s.n_result, last - first)
        for slot in range(s.n_result):
            assert shard[slot].tobytes() == planes[slot, first:last].tobytes(), (config, algorithm, slot)
        assert np.array_equal(rank["word"], words[first:last])
    s.close()


def _rccl():
    """The RCCL torch loaded (backend "nccl" is RCCL on ROCm): one library, one set of symbols."""
    import torch
    directory = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("librccl.so", "librccl.so.1"):
        path = os.path.join(directory, name)
        if os.path.exists(path):
            return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL("librccl.so", mode=ctypes.RTLD_GLOBAL)


def test_stats_allreduce_on_a_one_rank_rccl_communicator():
    """phq_stats_allreduce (ncclAllReduce of uint64 words, in place, on the caller's stream): one rank, so the buffer
    must come back unchanged — which still proves the symbol lookup, the enums and the in place call."""
    import torch

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    rccl = _rccl()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    unique = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(unique)) == 0
    communicator = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(communicator), 1, unique, 0) == 0
    try:
        job, recipe = synthetic.config_job("C2")
        n = 50000
        code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
        s = DecoderSet(job, device=0)
        s.stats_reset()
        base = torch.from_numpy(s.planes_from_rows(code, offsets)).cuda()
        qual = torch.from_numpy(s.planes_from_rows(quality, offsets)).cuda()
        flags = torch.from_numpy(flag).cuda()
        result = torch.zeros(s.n_result * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        word = torch.zeros(n, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        s.classify_device(n, 0, n, base, qual, flags, result, word, stream)
        torch.cuda.synchronize()
        before = s.stats_export()
        assert int(before[0] + before[1]) == n
        s.stats_allreduce(communicator.value, stream)
        torch.cuda.synchronize()
        assert np.array_equal(s.stats_export(), before)
        s.close()
    finally:
        rccl.ncclCommDestroy(communicator)


def test_stats_allreduce_on_a_two_rank_rccl_communicator():
    """phq_stats_allreduce with more than one rank, without torch.distributed: two decoder sets on two devices of this
    process, each with half of the reads, one RCCL communicator per device (ncclCommInitAll), the two calls inside one
    ncclGroupStart / ncclGroupEnd.  Both buffers must then hold, word for word, what one set counts over all the reads.
    Needs two GPUs: skipped on a one GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    rccl = _rccl()
    rccl.ncclCommInitAll.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    communicators = (ctypes.c_void_p * 2)()
    devices = (ctypes.c_int * 2)(0, 1)
    assert rccl.ncclCommInitAll(communicators, 2, devices) == 0
    try:
        job, recipe = synthetic.config_job("C5")
        n = 60000
        code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])

        def count(device, first, last):
            s = DecoderSet(job, device=device)
            s.stats_reset()
            torch.cuda.set_device(device)
            where = "cuda:%d" % device
            m = last - first
            base = torch.from_numpy(s.planes_from_rows(code[first:last], offsets)).to(where)
            qual = torch.from_numpy(s.planes_from_rows(quality[first:last], offsets)).to(where)
            flags = torch.from_numpy(flag[first:last].copy()).to(where)
            result = torch.zeros(max(1, s.n_result) * m * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=where)
            word = torch.zeros(m, dtype=torch.int32, device=where)
            s.classify_device(m, 0, m, base, qual, flags, result, word, torch.cuda.current_stream(device).cuda_stream)
            torch.cuda.synchronize(device)
            return s

        whole = count(0, 0, n)
        expected = whole.stats_export()
        halves = [count(0, 0, n // 2), count(1, n // 2, n)]
        assert rccl.ncclGroupStart() == 0
        for rank, s in enumerate(halves):
            torch.cuda.set_device(rank)
            s.stats_allreduce(communicators[rank], torch.cuda.current_stream(rank).cuda_stream)
        assert rccl.ncclGroupEnd() == 0
        for rank, s in enumerate(halves):
            torch.cuda.synchronize(rank)
            assert np.array_equal(s.stats_export(), expected), rank
        for s in halves + [whole]:
            s.close()
    finally:
        for communicator in communicators:
            rccl.ncclCommDestroy(communicator)
        torch.cuda.set_device(0)
