"""The FASTQ -> SAM leg of bench.py alone, for tuning the host side: usage fastq_leg.py N [target ...]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pheniqs_amd import DecoderSet, synthetic

n = int(sys.argv[1])
targets = sys.argv[2:] or ["file", "/dev/zero", "/dev/null"]
job, recipe = synthetic.config_job("C2")
code, quality, flag, _ = synthetic.generate_rows(job, min(n, 1 << 20), recipe["seed"])
spec = synthetic.config_spec("C2")
spec["template"] = {"transform": {"token": ["0::"]}}
s = DecoderSet(spec, device=0)
rng = np.random.default_rng(7)
letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
reps = -(-n // code.shape[0])
index_text = letters[np.tile(code, (reps, 1))[:n]]
index_quality = (np.minimum(np.tile(quality, (reps, 1))[:n], 93) + 33).astype(np.uint8)
template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 50), dtype=np.uint8)]
filtered = np.tile(flag, reps)[:n]
with tempfile.TemporaryDirectory(dir="/dev/shm") as directory:
    paths = []
    for k, (text, qual) in enumerate(((template, np.full((n, 50), ord("F"), dtype=np.uint8)), (index_text, index_quality))):
        path = os.path.join(directory, "segment_%d.fastq" % k)
        bench.write_fastq(path, k + 1, text, qual, filtered)
        paths.append(path)
    for target in targets:
        out = os.path.join(directory, "out.sam") if target == "file" else target
        best = 1e9
        for _ in range(3):
            if os.path.isfile(out):
                os.unlink(out)
            t0 = time.perf_counter()
            s.transcode_fastq_to_sam(paths, out, 1 << 20)
            best = min(best, time.perf_counter() - t0)
        print("%s: %.1f Mreads/s (%.3f s)" % (target, n / best / 1e6, best), flush=True)
