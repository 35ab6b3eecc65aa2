"""Model of the streaming kernel's pass A certificate on a single index set of 96 words of L nt (the benchmark's quality mix):
the share of reads it defers by the number of listed barcodes, with and without the H[n - k] term of the bound, and what
reaching two mismatches would buy.   usage: list_size.py L"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pheniqs_amd import synthetic, capi
L = int(sys.argv[1]); n = 200000
rng = np.random.default_rng(5)
words_s = synthetic.barcode_set(rng, 96, L)
spec = {"input segment cardinality": 1, "sample": synthetic.decoder_spec("pamld", [words_s], ["0::%d" % L])}
job = capi.compile_job(spec)
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
print("nearest neighbour distance histogram", np.bincount(np.sort(bd, axis=1)[:, 1]))
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
for LIST in (16, 12, 8, 6, 4):
    listed = order[:, :LIST]
    inlist = np.zeros((N, N), bool); inlist[np.arange(N)[:, None], listed] = True
    others = np.where(inlist[c], rel, 0).sum(axis=1) - winner
    true_rest = np.where(inlist[c], 0.0, rel).sum(axis=1)
    w = np.zeros((N, 4))
    for b in range(N):
        for o in range(N):
            if not inlist[b, o]: w[b, min(max(bd[b, o], 3), 6) - 3] += 1
    close = np.array([(~inlist[b] & (bd[b] < 3)).any() for b in range(N)])
    rest = boost * sum(w[c, i] * np.minimum(H[:, 2 + i], H[:, 3 + i] * inv) for i in range(4))
    rest = np.where(close[c], np.inf, rest)
    x_listed = others * inv; x_rest = rest * inv; x_low = x_listed + nop / (A * winner)
    fin = strict & (k <= 1) & (x_listed <= 5e-4) & (x_low + x_rest <= 5e-4) & (x_rest <= 1e-8) & (A * winner > 0.25 ** L)
    assert (true_rest[fin] <= rest[fin] * (1 + 1e-9)).all()
    print("L %d LIST %2d: deferred %.3f %%  (k=0: %.3f, k=1: %.3f, k>=2: %.3f of all)" % (L, LIST, 100 * (1 - fin.mean()), 100 * ((k == 0) & ~fin).mean(), 100 * ((k == 1) & ~fin).mean(), 100 * (k >= 2).mean()))
    xt = true_rest * inv
    fint = strict & (k <= 1) & (x_listed <= 5e-4) & (x_low + xt <= 5e-4) & (xt <= 1e-8) & (A * winner > 0.25 ** L)
    print("      with the TRUE rest: deferred %.3f %%  (k=0: %.3f, k=1: %.3f); reasons among k<=1 strict: listed too heavy %.3f, rest too heavy %.3f, noise floor %.3f" % (100 * (1 - fint.mean()), 100 * ((k == 0) & ~fint).mean(), 100 * ((k == 1) & ~fint).mean(),
          100 * (strict & (k <= 1) & (x_listed > 5e-4)).mean(), 100 * (strict & (k <= 1) & (x_listed <= 5e-4) & (xt > 1e-8)).mean(), 100 * (strict & (k <= 1) & (A * winner <= 0.25 ** L)).mean()))
print("--- k <= 2, bound = boost * sum w_n min(H[n-k], H[n] / winner)")
for LIST in (16, 8):
    listed = order[:, :LIST]
    inlist = np.zeros((N, N), bool); inlist[np.arange(N)[:, None], listed] = True
    others = np.where(inlist[c], rel, 0).sum(axis=1) - winner
    true_rest = np.where(inlist[c], 0.0, rel).sum(axis=1)
    w = np.zeros((N, 4))
    for b in range(N):
        for o in range(N):
            if not inlist[b, o]: w[b, min(max(bd[b, o], 3), 6) - 3] += 1
    close = np.array([(~inlist[b] & (bd[b] < 3)).any() for b in range(N)])
    # M-aware boost: product over mismatching positions of max(rho, 1)
    Mmask = mis[np.arange(n), c]
    boostM = np.where(Mmask, np.maximum(rho, 1.0), 1.0).prod(axis=1)
    for variant in ("min", "nomin"):
        kk = np.minimum(k, 2)
        if variant == "min":
            rest = boostM * sum(w[c, i] * np.minimum(H[np.arange(n), np.maximum(3 + i - kk, 0)], H[:, 3 + i] * inv) for i in range(4))
        else:
            rest = boostM * sum(w[c, i] * H[:, 3 + i] * inv for i in range(4))
        rest = np.where(close[c], np.inf, rest)
        x_listed = others * inv; x_rest = rest * inv; x_low = x_listed + nop / (A * winner)
        for kmax in (1, 2):
            fin = strict & (k <= kmax) & (x_listed <= 5e-4) & (x_low + x_rest <= 5e-4) & (x_rest <= 1e-8) & (A * winner > 0.25 ** L)
            bad = fin & (true_rest > rest * (1 + 1e-9))
            print("L %d LIST %2d %-5s k<=%d: deferred %.3f %%  (k=0: %.3f, k=1: %.3f, k=2: %.3f, k>2: %.3f)  violations %d" % (L, LIST, variant, kmax, 100 * (1 - fin.mean()), 100 * ((k == 0) & ~fin).mean(), 100 * ((k == 1) & ~fin).mean(), 100 * ((k == 2) & ~fin).mean(), 100 * (k > 2).mean(), bad.sum()))
