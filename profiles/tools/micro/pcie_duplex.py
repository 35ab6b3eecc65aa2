"""Does this box move bytes both ways over PCIe at once?  One GiB host -> device and one GiB device -> host, each alone and
then together on two streams (pinned host memory, HIP events).  usage: python profiles/tools/micro/pcie_duplex.py"""
import torch

n = 1 << 30
host_in = torch.empty(n, dtype=torch.uint8).pin_memory()
host_out = torch.empty(n, dtype=torch.uint8).pin_memory()
device_in = torch.empty(n, dtype=torch.uint8, device="cuda")
device_out = torch.ones(n, dtype=torch.uint8, device="cuda")
up, down = torch.cuda.Stream(), torch.cuda.Stream()


def timed(run):
    torch.cuda.synchronize()
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    begin.record()
    run()
    torch.cuda.synchronize()
    end.record()
    torch.cuda.synchronize()
    return begin.elapsed_time(end) * 1e-3


def h2d():
    with torch.cuda.stream(up):
        device_in.copy_(host_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(down):
        host_out.copy_(device_out, non_blocking=True)


for _ in range(2):
    h2d(); d2h()
torch.cuda.synchronize()
import time
for name, run, moved in (("host -> device", h2d, n), ("device -> host", d2h, n), ("both at once", lambda: (h2d(), d2h()), 2 * n)):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-16s %.1f GB/s" % (name, moved / dt / 1e9))
