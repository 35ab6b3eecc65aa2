"""Model of the list tiers' certificate on S384 (384 single indexes of 8 nt, the benchmark's quality mix): the share of reads deferred by
the number of listed barcodes, with the bound and with the TRUE weight of what is not listed.   usage: list_size_384.py
16 listed: 22.7 % deferred (17.9 % even with the true weight); 48: 11.8 %; 96: 8.9 %; the floor (noise, N, two mismatches, heavy lists) 8.3 %."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))))
from pheniqs_amd import synthetic, capi
L = 8; n = 60000
job, recipe = synthetic.config_job("S384", "pamld")
dec = job["sample"]; words, prior = synthetic._codec_arrays(dec); N = words.shape[0]
code, quality, flag, offsets = synthetic.generate_rows(job, n, 11)
bam2code = np.full(16, 255, dtype=np.uint8); bam2code[[1, 2, 4, 8]] = [0, 1, 2, 3]
obs = bam2code[code]; q = quality.astype(np.int64); strict = (obs != 255).all(axis=1)
fpp = (10.0 ** -0.1) ** np.arange(128.0); tpp = 1 - fpp
fm = np.where(q > 0, tpp[q], 1.0); fe = np.where(q > 0, fpp[q], 1.0); rho = fe / fm
noise = float(dec["noise"]); pri = np.array([float(b["concentration"]) for b in dec["codec"].values()]); nop = noise * 0.25 ** L / pri[0]
mis = obs[:, None, :] != words[None, :, :]
dist = mis.sum(axis=2); c = dist.argmin(axis=1); k = dist[np.arange(n), c]
bd = (words[:, None, :] != words[None, :, :]).sum(axis=2)
rel = np.where(mis, rho[:, None, :], 1.0).prod(axis=2)
winner = rel[np.arange(n), c]
srt = -np.sort(-rho, axis=1); cum = np.cumprod(srt, axis=1)
H = np.ones((n, L + 2))
for m in range(1, L + 1): H[:, m] = cum[:, m - 1]
H[:, L + 1] = 0
for m in range(L - 1, -1, -1): H[:, m] = np.maximum(H[:, m], H[:, m + 1])
order = np.argsort(bd, axis=1, kind="stable")
inv = 1 / winner; boost = np.maximum(winner, 1)
A = fm.prod(axis=1)
print("barcodes at distance d of a barcode (mean):", [(d, round(float((bd == d).sum(axis=1).mean()), 1)) for d in range(1, 8)])
for LIST in (16, 24, 32, 48, 64, 96):
    listed = order[:, :LIST]
    inlist = np.zeros((N, N), bool); inlist[np.arange(N)[:, None], listed] = True
    others = np.where(inlist[c], rel, 0).sum(axis=1) - winner
    true_rest = np.where(inlist[c], 0.0, rel).sum(axis=1)
    w = np.zeros((N, 4))
    for d in range(3, 9):
        w[:, min(d, 6) - 3] += ((bd == d) & ~inlist).sum(axis=1)
    close = ((bd < 3) & ~inlist).any(axis=1)
    rest = boost * sum(w[c, i] * np.minimum(H[:, 2 + i], H[:, 3 + i] * inv) for i in range(4))
    rest = np.where(close[c], np.inf, rest)
    x_listed = others * inv; x_rest = rest * inv; x_low = x_listed + nop / (A * winner)
    fin = strict & (k <= 1) & (x_listed <= 5e-4) & (x_low + x_rest <= 5e-4) & (x_rest <= 9e-9) & (A * winner > 0.25 ** L)
    fin2 = strict & (k <= 1) & (x_listed <= 5e-4) & (x_low + true_rest * inv <= 5e-4) & (true_rest * inv <= 9e-9) & (A * winner > 0.25 ** L)
    print("LIST %2d: deferred %.2f %%  (listed too heavy %.2f, bound %.2f; with the true rest %.2f)" % (LIST, 100 * (1 - fin.mean()), 100 * (strict & (k <= 1) & (x_listed > 5e-4)).mean(),
          100 * (strict & (k <= 1) & (x_listed <= 5e-4) & ~fin).mean(), 100 * (1 - fin2.mean())))
