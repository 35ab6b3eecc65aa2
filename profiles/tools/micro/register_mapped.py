"""Can a read only mapping of a file be registered with the device (hipHostRegister) and copied from at the link's rate?
What it costs to register, against reading the file into a pinned buffer first.  usage: python register_mapped.py [MiB]"""
import ctypes, mmap, os, sys, tempfile, time
import numpy as np
import torch

size = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
directory = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.NamedTemporaryFile(dir=directory) as handle:
    handle.write(np.random.default_rng(1).integers(0, 255, size=size, dtype=np.uint8).tobytes())
    handle.flush()
    descriptor = os.open(handle.name, os.O_RDONLY)
    mapped = mmap.mmap(descriptor, size, prot=mmap.PROT_READ, flags=mmap.MAP_PRIVATE)
    array = np.frombuffer(mapped, dtype=np.uint8)
    pointer = array.ctypes.data
    device = torch.empty(size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    runtime = torch.cuda.cudart()
    t0 = time.perf_counter()
    status = runtime.cudaHostRegister(pointer, size, 0)
    t1 = time.perf_counter()
    print("hipHostRegister of a %d MiB read only file mapping: status %s, %.1f ms" % (size >> 20, status, (t1 - t0) * 1e3))
    if int(status) == 0:
        lib = ctypes.CDLL("libamdhip64.so")
        lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        for _ in range(3):
            t0 = time.perf_counter()
            assert lib.hipMemcpyAsync(device.data_ptr(), pointer, size, 1, None) == 0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("copy from the registered mapping: %.1f GB/s" % (size / dt / 1e9))
        assert bool((device.cpu().numpy() == array).all())
        t0 = time.perf_counter()
        runtime.cudaHostUnregister(pointer)
        print("unregister: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    pinned = torch.empty(size, dtype=torch.uint8).pin_memory()
    t0 = time.perf_counter()
    got = os.preadv(descriptor, [memoryview(pinned.numpy())], 0)
    t1 = time.perf_counter()
    print("pread into a pinned buffer (one thread): %.1f GB/s" % (got / (t1 - t0) / 1e9))
