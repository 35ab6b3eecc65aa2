import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
seed = int(sys.argv[1])
import test_gpu_random_jobs as t
import test_gpu_parity as p
import jobspec
from pheniqs_amd import synthetic, capi
import torch
# build the job exactly as the test does
rng = np.random.default_rng(1000 + seed)
algorithm = "pamld" if seed % 2 == 0 else "mdd"
sample, n_input = t._random_decoder(rng, algorithm, "sample")
spec = {"input segment cardinality": n_input, "sample": sample}
if rng.random() < 0.5:
    cellular, needed = t._random_decoder(rng, str(rng.choice(["pamld", "mdd"])), "cellular")
    spec["cellular"] = [cellular]
    spec["input segment cardinality"] = max(n_input, needed)
if rng.random() < 0.3:
    spec["molecular"] = [{"algorithm": "naive", "transform": {"token": ["0::4"]}}]
job = jobspec.compile_job(spec)
for name, d in jobspec.iter_decoders(job):
    codec = d.get("codec") or {}
    print(name, d.get("algorithm"), "barcodes", len(codec), "lengths", d.get("barcode length"), "threshold", d.get("confidence threshold"), "noise", d.get("noise"), "hq", d.get("high quality threshold"), d.get("high quality distance threshold"),
          "priors differ", len(set(float(b["concentration"]) for b in codec.values())) > 1)
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
from oracle import pyoracle as oracle_module; oracle_module.build()
engine = oracle_module.OracleJob(job)
expected, expected_word = engine.classify(code, quality, flag)
for mode in (capi.MODE_LITERAL, capi.MODE_FAST):
    s, got, word, words = p.run_gpu(torch, job, code, quality, flag, offsets, mode=mode, pieces=pieces)
    for slot in range(s.n_result):
        delta = np.abs(got[slot]["confidence"] - expected[:, slot]["confidence"])
        worst = int(delta.argmax())
        print("mode", mode, "slot", slot, "max delta", delta.max(), "at", worst, "got", got[slot][worst], "expected", expected[worst, slot])
        bad = np.nonzero(delta > 1e-8)[0]
        for i in bad[:5]:
            print("   read", i, "code", code[i], "quality", quality[i], "got", got[slot][i], "expected", expected[i, slot])
    count, pf_count = engine.outgoing()
    print("mode", mode, "outgoing mine", int(words[0] + words[1]), int(words[1]), "oracle", count, pf_count, "pieces", pieces)
    for k in range(s.n_decoder):
        mine, theirs = s.stats_options(words, k), engine.options(k)
        for name in ("count", "pf_count"):
            if not np.array_equal(mine[name], theirs[name]):
                print("   decoder", k, name, "mine", mine[name][:6], "oracle", theirs[name][:6])
