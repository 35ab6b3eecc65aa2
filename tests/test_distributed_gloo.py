"""The N > 1 path on CPU: two processes, gloo backend.  Each rank decodes its contiguous shard (with the CPU
oracle standing in for the kernels), encodes its counters in the library's integer layout, and the ranks merge
with one SUM all-reduce — the merged buffer must equal the single process buffer word for word, and the
report written from it must be identical on every rank."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _words(s, engine):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_capi_cpu import _words_from_oracle
    return _words_from_oracle(s, engine)


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import pyoracle
    from pheniqs_amd import DecoderSet, capi, distributed, synthetic
    r, _, w = distributed.initialise("gloo")
    assert (r, w) == (rank, world)
    job, recipe = synthetic.config_job("C5")
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    first, last = distributed.shard_range(n, rank, world)
    engine = pyoracle.OracleJob(job)
    engine.classify(code[first:last], quality[first:last], flag[first:last])
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    words = _words(s, engine)
    tensor = torch.from_numpy(words.view(np.int64).copy())
    distributed.allreduce_counters(tensor)
    merged = tensor.numpy().view(np.uint64)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), merged)
    with open(os.path.join(out_dir, "report_%d.json" % rank), "w") as handle:
        handle.write(s.report(merged, (n, int(n - flag.sum()))))
    dist.destroy_process_group()


def test_two_rank_counter_merge_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle
    from pheniqs_amd import DecoderSet, capi, distributed, synthetic
    n, world = 20001, 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)

    job, recipe = synthetic.config_job("C5")
    code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
    engine = pyoracle.OracleJob(job)
    engine.classify(code, quality, flag)
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    single = _words(s, engine)
    merged = [np.load(os.path.join(tmp_path, "merged_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(merged[0], merged[1])
    # integer words are exact; the confidence limbs were rounded per rank from the oracle's double sums
    exact = np.ones(s.stats_words, dtype=bool)
    for d in s.decoders:
        for b in range(d["n_barcode"] + 1):
            for half in (0, 6):
                exact[d["stats_offset"] + b * 12 + half + 2] = False
                exact[d["stats_offset"] + b * 12 + half + 3] = False
    assert np.array_equal(merged[0][exact], single[exact])
    for k in range(s.n_decoder):
        a, b = s.stats_options(merged[0], k), s.stats_options(single, k)
        assert np.allclose(a["accumulated_confidence"], b["accumulated_confidence"], rtol=1e-13, atol=1e-12)
    reports = [open(os.path.join(tmp_path, "report_%d.json" % r)).read() for r in range(world)]
    assert reports[0] == reports[1]
    report = json.loads(reports[0])
    assert report["outgoing"]["count"] == n and report["incoming"]["count"] == n
    assert [distributed.shard_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
