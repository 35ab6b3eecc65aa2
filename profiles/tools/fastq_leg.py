"""The FASTQ -> SAM legs of bench.py alone: python fastq_leg.py READS CAPACITY [target ...] (targets: file zero null); three runs each,
with what phq_device_path_counters says the run did on the device."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pheniqs_amd import DecoderSet, capi, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
capacity = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
targets = sys.argv[3:] or ["file", "zero", "null"]
spec = synthetic.config_spec("C2", "pamld")
spec["template"] = {"transform": {"token": ["0::"]}}
if os.environ.get("QC"):
    spec["enable quality control"] = True
SPLIT = os.environ.get("SPLIT")                  # a FASTQ file per barcode (SPLIT=plain) or a gzip one (SPLIT=gz): target "split"
s = DecoderSet(spec, device=0)
if os.environ.get("MODE"):                       # fast | literal; default: PHQ_MODE_AUTO (exact tags where SAM records are written)
    s.set_mode({"fast": capi.MODE_FAST, "literal": capi.MODE_LITERAL}[os.environ["MODE"]])
job = capi.compile_job(spec)
pool = 1 << 20
code, quality, flag, _ = synthetic.generate_rows(job, pool, 3)
rng = np.random.default_rng(7)
letters = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
reps = -(-n // pool)
index_text = letters[np.tile(code, (reps, 1))[:n]]
index_quality = (np.minimum(np.tile(quality, (reps, 1))[:n], 93) + 33).astype(np.uint8)
template = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 50), dtype=np.uint8)]
template_quality = np.full((n, 50), ord("F"), dtype=np.uint8)
filtered = np.tile(flag, reps)[:n]
with tempfile.TemporaryDirectory(dir="/dev/shm") as directory:
    paths = []
    for k, (text, qual) in enumerate(((template, template_quality), (index_text, index_quality))):
        path = os.path.join(directory, "segment_%d.fastq" % k)
        bench.write_fastq(path, k + 1, text, qual, filtered)
        if os.environ.get("GZ"):                      # the inputs as gzip streams
            import gzip
            with open(path, "rb") as plain, gzip.open(path + ".gz", "wb", compresslevel=int(os.environ["GZ"])) as packed:
                packed.write(plain.read())
            os.unlink(path)
            path += ".gz"
        paths.append(path)
    for name in targets:
        if name == "split":
            split_spec = dict(spec)
            split_spec["input"] = paths
            split_spec["output"] = [os.path.join(directory, "rest.fastq")]
            split_spec["sample"] = dict(spec["sample"])
            split_spec["sample"]["codec"] = {key: dict(value, output=[os.path.join(directory, "%d.fastq%s" % (k, ".gz" if SPLIT == "gz" else ""))])
                                             for k, (key, value) in enumerate(spec["sample"]["codec"].items())}
            split = DecoderSet(split_spec, device=0)
            for attempt in range(3):
                before = capi.device_path_counters()
                t0 = time.perf_counter()
                incoming, _ = split.transcode_job(capacity)
                dt = time.perf_counter() - t0
                after = capi.device_path_counters()
                print(name, attempt, "%.1f Mreads/s  %.1f ms" % (n / dt / 1e6, dt * 1e3), {k: after[k] - before[k] for k in after}, flush=True)
            split.close()
            continue
        target = {"file": os.path.join(directory, "out.sam"), "zero": "/dev/zero", "null": "/dev/null"}[name]
        for attempt in range(3):
            if os.path.isfile(target):
                os.unlink(target)
            before = capi.device_path_counters()
            t0 = time.perf_counter()
            incoming, _ = s.transcode_fastq_to_sam(paths, target, capacity)
            dt = time.perf_counter() - t0
            after = capi.device_path_counters()
            print(name, attempt, "%.1f Mreads/s  %.1f ms" % (n / dt / 1e6, dt * 1e3), {k: after[k] - before[k] for k in after}, flush=True)
