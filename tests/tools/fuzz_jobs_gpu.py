"""Damaged job documents through phq_create and a small classification on the GPU: an error code or a result, never
a crash.  Run by hand on a GPU box (bounded: `timeout 600 python tests/tools/fuzz_jobs_gpu.py 400`)."""
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bdggg                                                     # noqa: E402
from pheniqs_amd import DecoderSet, PheniqsError, synthetic      # noqa: E402


def damaged(text, rng):
    for _ in range(rng.randint(1, 3)):
        at = rng.randrange(len(text))
        kind = rng.randrange(6)
        if kind == 0:
            text = text[:at] + chr(rng.randrange(32, 127)) + text[at + 1:]
        elif kind == 1:
            text = text[:at] + text[at + rng.randint(1, 8):]
        elif kind == 2:
            text = text[:at] + text[at:at + rng.randint(1, 16)] + text[at:]
        elif kind == 3:
            text = text[:at] + rng.choice(list('{}[],:"0123456789-')) + text[at + 1:]
        elif kind == 4:
            text = text[:at] + "null" + text[at + 4:]
        else:                                                   # a number made huge or negative
            digits = [i for i, c in enumerate(text) if c.isdigit()]
            if digits:
                at = rng.choice(digits)
                text = text[:at] + rng.choice(["-1", "0", "255", "4096", "1e9", "1e-300"]) + text[at + 1:]
    return text


def main():
    import torch  # noqa: F401  (first, so that the library shares torch's HIP runtime)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(7)
    sources = [json.dumps(bdggg.compiled_reference_job()), json.dumps(bdggg.raw_job())]
    sources += [json.dumps(synthetic.config_spec(name, algorithm)) for name in ("C2", "C3", "C4", "C5") for algorithm in ("pamld", "mdd")]
    created = refused = 0
    for k in range(rounds):
        text = damaged(rng.choice(sources), rng)
        try:
            s = DecoderSet(text, device=0)
        except PheniqsError:
            refused += 1
            continue
        created += 1
        n = 300
        width = max(int(s.plane_width), 4)
        code = np.random.default_rng(k).integers(0, 16, size=n * width, dtype=np.uint8)
        quality = np.random.default_rng(k + 1).integers(0, 256, size=n * width, dtype=np.uint8)
        try:
            s.classify_host(n, code, quality, None)
            s.report(s.stats_export(), (n, n))
        except PheniqsError:
            pass
        s.close()
    print("fuzz: %d documents, %d refused, %d created and classified" % (rounds, refused, created))

    # damaged FASTQ text through the transcoder: a sequence / IO error or records, never a crash
    import tempfile
    job = json.dumps(bdggg.compiled_reference_job())
    originals = [open(os.path.join(bdggg.HERE, "BDGGG_s0%d.fastq" % k), "rb").read().decode("latin-1") for k in (1, 2, 3)]
    finished = failed = 0
    with tempfile.TemporaryDirectory() as directory:
        for k in range(max(rounds // 10, 20)):
            paths = []
            victim = rng.randrange(3)
            for f, text in enumerate(originals):
                if f == victim:
                    text = damaged(text, rng)
                    if rng.random() < 0.3:
                        text = text[:rng.randrange(len(text))]                    # truncated
                path = os.path.join(directory, "s%d.fastq" % f)
                open(path, "wb").write(text.encode("latin-1"))
                paths.append(path)
            s = DecoderSet(job, device=0)
            try:
                s.transcode_fastq_to_sam(paths, os.path.join(directory, "out.sam"), rng.choice([7, 64, 1 << 16]))
                finished += 1
            except PheniqsError:
                failed += 1
            s.close()
    print("fuzz: FASTQ files, %d runs finished, %d stopped with an error" % (finished, failed))


if __name__ == "__main__":
    main()
