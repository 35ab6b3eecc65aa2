"""Two (or more) ranks on ONE GPU, kernels in sharded mode: every rank classifies its contiguous shard of the same
seeded batch with phq_classify_device and the ranks merge the DEVICE counter buffer with one SUM all-reduce
(gloo: both ranks share cuda:0, RCCL wants one device per rank).  Writes per rank result planes, status words and
the merged counter words to OUT for tests/test_gpu_sharded.py, which holds them against the unsharded run.

Started by tests/conftest.py at the beginning of a `-m gpu` session — before the pytest process has touched the
GPU — and spawns its ranks before touching it itself.   usage: sharded_ranks.py OUT N WORLD
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = (("C2", "pamld"), ("C5", "pamld"), ("C2", "mdd"))


def worker(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from pheniqs_amd import DecoderSet, RESULT_DTYPE, distributed, synthetic
    r, _, w = distributed.initialise("gloo")
    assert (r, w) == (rank, world)
    torch.cuda.set_device(0)
    for config, algorithm in CASES:
        job, recipe = synthetic.config_job(config, algorithm)
        code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
        first, last = distributed.shard_range(n, rank, world)
        m = last - first
        s = DecoderSet(job, device=0)
        s.stats_reset()
        base = torch.from_numpy(s.planes_from_rows(code[first:last], offsets)).cuda()
        qual = torch.from_numpy(s.planes_from_rows(quality[first:last], offsets)).cuda()
        flags = torch.from_numpy(np.ascontiguousarray(flag[first:last])).cuda()
        result = torch.zeros(max(1, s.n_result * m) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        word = torch.zeros(max(1, m), dtype=torch.int32, device="cuda")
        s.classify_device(m, 0, m, base, qual, flags, result, word, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        own = s.stats_export()
        counters = s.stats_tensor()                      # the device buffer itself
        distributed.allreduce_counters(counters)
        torch.cuda.synchronize()
        merged = s.stats_export()
        np.savez(os.path.join(out, "%s_%s_rank%d.npz" % (config, algorithm, rank)), first=first, last=last, own=own, merged=merged,
                 result=result.cpu().numpy(), word=word.cpu().numpy())
        s.close()
    dist.destroy_process_group()


def main():
    out, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.makedirs(out, exist_ok=True)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(world, port, n, out), nprocs=world, join=True)
    open(os.path.join(out, "done"), "w").write("ok\n")


if __name__ == "__main__":
    main()
