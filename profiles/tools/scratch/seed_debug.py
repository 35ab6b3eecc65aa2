"""one seed of tests/test_gpu_random_jobs.py::test_random_job, with what differs printed: python profiles/tools/scratch/seed_debug.py SEED [mode]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jobspec
from pheniqs_amd import synthetic, capi
import test_gpu_random_jobs as T
import test_gpu_parity as P

seed = int(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else capi.MODE_LITERAL
rng = np.random.default_rng(1000 + seed)
algorithm = "pamld" if seed % 2 == 0 else "mdd"
sample, n_input = T._random_decoder(rng, algorithm, "sample")
spec = {"input segment cardinality": n_input, "sample": sample}
if rng.random() < 0.5:
    cellular, needed = T._random_decoder(rng, str(rng.choice(["pamld", "mdd"])), "cellular")
    spec["cellular"] = [cellular]
    spec["input segment cardinality"] = max(n_input, needed)
if rng.random() < 0.3:
    spec["molecular"] = [{"algorithm": "naive", "transform": {"token": ["0::4"]}}]
job = jobspec.compile_job(spec)
for name, d in jobspec.iter_decoders(job):
    codec = d.get("codec") or {}
    print(name, d.get("algorithm"), len(codec), "barcodes", [len(w) for w in next(iter(codec.values()))["barcode"]] if codec else None,
          "priors differ" if len({r["concentration"] for r in codec.values()}) > 1 else "uniform", {k: d[k] for k in d if k in ("confidence threshold", "noise", "distance tolerance")})
largest = max(len(d.get("codec") or {}) for _, d in jobspec.iter_decoders(job))
n = 6000 if largest <= 150 else 2500
code, quality, flag, offsets = synthetic.generate_rows(job, n, seed)
width = code.shape[1]
noisy = rng.random((n, width)) < 0.03
code[noisy] = np.array([15, 0, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=int(noisy.sum()))]
quality[rng.random((n, width)) < 0.03] = 0
wild = rng.random((n, width)) < 0.01
quality[wild] = rng.integers(0, 256, size=int(wild.sum())).astype(np.uint8)
quality[: n // 10] = int(rng.choice([2, 12, 30, 40]))
for i in range(min(width, 40)):
    code[n // 2 + i, i:] = 0
    quality[n // 2 + i, i:] = 0
pieces = int(rng.integers(1, 4))
from oracle import pyoracle
pyoracle.build()
engine = pyoracle.OracleJob(job)
expected, expected_word = engine.classify(code, quality, flag)
s, got, word, words = P.run_gpu(torch, job, code, quality, flag, offsets, mode=mode, pieces=pieces)
print("chain", [c["kernel"] for c in s.plan_chain(n, mode)])
for slot in range(s.n_result):
    bad = np.nonzero((got[slot]["index"] != expected[:, slot]["index"]) | (got[slot]["status"] != expected[:, slot]["status"]))[0]
    print("slot", slot, "differs on", bad.size, "reads")
    decoders = [d for _, d in jobspec.iter_decoders(job) if d.get("codec")]
    d = [d for _, d in jobspec.iter_decoders(job) if d.get("codec")][slot] if slot < len([1 for _, d in jobspec.iter_decoders(job) if d.get("codec")]) else None
    names = list(d["codec"].keys()) if d else []
    for i in bad[:6]:
        gi, ei = int(got[slot][i]["index"]), int(expected[i, slot]["index"])
        print(i, "got", got[slot][i], "expected", expected[i, slot], "code", code[i].tolist(), "quality", quality[i].tolist())
        if d:
            byindex = {r.get("index", k + 1): (name, r["barcode"]) for k, (name, r) in enumerate(d["codec"].items())}
            print("   got barcode", byindex.get(gi), "expected barcode", byindex.get(ei))
