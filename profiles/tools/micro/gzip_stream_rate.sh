#!/bin/bash
# a FASTQ file like the bench's template file (50 nt reads, Illumina names), gzip level 6, then the rates by thread count
set -e
cd "$(dirname "$0")"
g++ -std=c++17 -O3 -o /tmp/gzip_stream_rate gzip_stream_rate.cpp -lz -lpthread
python3 - <<'PY'
import numpy as np, zlib
rng = np.random.default_rng(3)
n = 1500000
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 50), dtype=np.uint8)]
quality = np.frombuffer(b"#-:F", dtype=np.uint8)[rng.choice(4, size=(n, 50), p=[.03, .07, .1, .8])]
lines = []
for k in range(n):
    lines.append(b"@A00123:45:HXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT\n" % (1000 + k % 30000, 1000 + k // 30))
    lines.append(bases[k].tobytes() + b"\n+\n" + quality[k].tobytes() + b"\n")
text = b"".join(lines)
packer = zlib.compressobj(6, zlib.DEFLATED, 31)
open("/tmp/rate.fastq.gz", "wb").write(packer.compress(text) + packer.flush())
print("text %d bytes" % len(text))
PY
ls -l /tmp/rate.fastq.gz
/tmp/gzip_stream_rate /tmp/rate.fastq.gz
