"""Numerical model of the paired barcode certificate (pamld_pair): which share of the reads of the second C4 / C4V
decoder it finishes, by cause, and what finer quality classes in the bound would buy.   usage: pair_certificate.py [C4|C4V] [n]"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from pheniqs_amd import synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
job, recipe = synthetic.config_job(name, "pamld")
decoders = [d for _, d in synthetic.iter_decoders(job)]
code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
which = [i for i, d in enumerate(decoders) if int(d["nucleotide cardinality"]) == 20][0]
dec = decoders[which]
L = 20
obs_code = code[:, offsets[which]:offsets[which] + L]
q = quality[:, offsets[which]:offsets[which] + L].astype(np.int64)
words, prior = synthetic._codec_arrays(dec)
N = words.shape[0]
bam2code = np.full(16, 255, dtype=np.uint8); bam2code[[1, 2, 4, 8]] = [0, 1, 2, 3]
obs = bam2code[obs_code]
strict = (obs != 255).all(axis=1)
e = 10.0 ** (-q / 10.0)
fm = np.where(q > 0, 1 - e, 1.0); fe = np.where(q > 0, e, 1.0)
mis = obs[:, None, :] != words[None, :, :]
dist = mis.sum(axis=2)
c = dist.argmin(axis=1)
k = dist[np.arange(n), c]
k0 = mis[np.arange(n), c, :10].sum(axis=1); k1 = mis[np.arange(n), c, 10:].sum(axis=1)
known = (k0 <= 1) & (k1 <= 1)
bd = (words[:, None, :] != words[None, :, :]).sum(axis=2) + np.eye(N, dtype=int) * 99
radius = bd.min(axis=1)
print(name, "N", N, "radius histogram", np.bincount(radius)[:14])
P = np.where(mis, fe[:, None, :], fm[:, None, :]).prod(axis=2)
Pc = P[np.arange(n), c]
A = fm.prod(axis=1)
true_rest = P.sum(axis=1) - Pc
noise = float(dec["noise"]) * 0.25 ** L / float(prior[0] / prior.sum() * (1 - float(dec["noise"])))

# the other barcodes of each barcode by distance: at its radius, one further away, all the rest (the kernel's three classes)
in_class = [(bd == radius[:, None]).sum(axis=1), (bd == radius[:, None] + 1).sum(axis=1), ((bd >= radius[:, None] + 2) & (bd < 99)).sum(axis=1)]


def bound(marks, wild_quality=4, classes=1):
    """marks: ascending qualities; positions below marks[0] are free, then rho(marks[i]) for those below marks[i + 1], rho(marks[-1]) for the rest.
    classes = 1: every other barcode charged at the radius; 3: those at the radius, one further, the rest two further"""
    rho = [10 ** (-m / 10) / (1 - 10 ** (-m / 10)) for m in marks]
    below = [(q < m).sum(axis=1) for m in marks]

    def charged(outside):
        needed = np.maximum(outside - below[0], 0)
        value = 4.0 ** (q < wild_quality).sum(axis=1)
        for i in range(len(marks) - 1):
            take = np.minimum(needed, below[i + 1] - below[i])
            value = value * rho[i] ** take
            needed = needed - take
        return value * rho[-1] ** needed

    if classes == 1:
        return (N - 1) * A * charged(radius[c] - k)
    return A * sum(in_class[j][c] * charged(radius[c] - k + j) for j in range(3))

known2 = (k0 <= 2) & (k1 <= 2)          # tables that name the barcode of a word up to TWO mismatches away
REST_MAX = 1e-8          # the kernel's (1e-9 when the first four lines below were the choice to make)
for marks, reach, classes in (([20, 30, 36], 1, 1), ([12, 20, 30, 36], 1, 1), ([12, 20, 25, 30, 36], 1, 1), ([12, 20, 30, 36], 2, 1), ([12, 20, 25, 30, 36], 2, 1),
                              ([12, 20, 25, 30, 36, 37], 2, 1), ([12, 20, 25, 30, 36], 2, 3)):
    U = bound(marks, classes=classes)
    named = known if reach == 1 else known2
    most = 2 if reach == 1 else 4
    assert (true_rest <= U * (1 + 1e-9))[named & strict & (k <= most)].all(), "bound violated"
    x_rest = U / Pc; x_low = noise / Pc
    finished = strict & named & (k <= most) & (Pc > 0.25 ** L) & (x_low + x_rest <= 3e-4) & (x_rest <= REST_MAX)
    print("marks", marks, "reach", reach, "classes", classes, "finished %.2f %%" % (100 * finished.mean()), " of k=0: %.2f %%, k=1: %.2f %%, k=2: %.2f %%" % tuple(100 * finished[k == kk].mean() for kk in (0, 1, 2)),
          " shares k=0 %.1f k=1 %.1f k=2 %.1f" % tuple(100 * (k == kk).mean() for kk in (0, 1, 2)))
U = true_rest
finished = strict & known & (k <= 2) & (Pc > 0.25 ** L) & ((noise + U) / Pc <= 3e-4)
print("exact remainder: finished %.2f %%" % (100 * finished.mean()))
