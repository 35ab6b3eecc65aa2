"""Greads/s of a single index decoder of COUNT barcodes of LENGTH nt: python time_shape.py LENGTH COUNT [READS] [pamld|mdd]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pheniqs_amd import DecoderSet, RESULT_DTYPE, capi, synthetic
length, count = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000_000
algorithm = sys.argv[4] if len(sys.argv) > 4 else "pamld"
rng = np.random.default_rng(5)
words = synthetic.barcode_set(rng, count, length)
spec = {"input segment cardinality": 1, "sample": synthetic.decoder_spec(algorithm, [words], ["0::%d" % length])}
job = capi.compile_job(spec)
s = DecoderSet(spec, device=0)
s.set_mode(capi.MODE_FAST)
base, qual, flag, host = synthetic.generate_planes_device(torch, job, s.decoders, s.plane_width, n, 3, "cuda")
result = torch.zeros(s.n_result * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
word = torch.zeros(n, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
s.profile_enable(True)
for _ in range(2):
    s.classify_device(n, 0, n, base, qual, flag, result, word, stream)
torch.cuda.synchronize()
s.profile_chain()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    s.classify_device(n, 0, n, base, qual, flag, result, word, stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
chain = s.profile_chain()
by = {}
for e in chain:
    by[e["kernel"]] = by.get(e["kernel"], 0) + e["ms"] / steps
print(algorithm, "%d x %d nt: %.2f Greads/s  %.3f ms per %d M reads" % (count, length, n / dt / 1e9, dt * 1e3, n // 1000000), {k: round(v, 3) for k, v in by.items()})
