"""Numerical model of the paired tier's bound (pamld_pair.hpp) by cause of deferral, with and without the read's own mismatch
positions among the positions the bound charges (round 5: without).   usage: pair_bound_own.py CONFIG [reads] [decoder]
U384: 86.6 -> 93.8 % of the reads finished (97.3 % with the true weight of the others); L12: 77.4 -> 86.3 % (97.3 %)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))))
from pheniqs_amd import synthetic
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
job, recipe = synthetic.config_job(name, "pamld")
decoders = [d for _, d in synthetic.iter_decoders(job)]
code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dec = decoders[which]
L = int(dec["nucleotide cardinality"])
obs_code = code[:, offsets[which]:offsets[which] + L]
q = quality[:, offsets[which]:offsets[which] + L].astype(np.int64)
words, prior = synthetic._codec_arrays(dec)
N = words.shape[0]
bam2code = np.full(16, 255, dtype=np.uint8); bam2code[[1, 2, 4, 8]] = [0, 1, 2, 3]
obs = bam2code[obs_code]
strict = (obs != 255).all(axis=1)
e = 10.0 ** (-q / 10.0)
fm = np.where(q > 0, 1 - e, 1.0); fe = np.where(q > 0, e, 1.0)
P = np.empty((n, N))
dist = np.empty((n, N), dtype=np.int64)
for s in range(0, n, 20000):
    mis = obs[s:s+20000, None, :] != words[None, :, :]
    dist[s:s+20000] = mis.sum(axis=2)
    P[s:s+20000] = np.where(mis, fe[s:s+20000, None, :], fm[s:s+20000, None, :]).prod(axis=2)
c = P.argmax(axis=1)
k = dist[np.arange(n), c]
Pc = P[np.arange(n), c]
rest = P.sum(axis=1) - Pc
noise = float(dec["noise"]); rbp = float(dec["random barcode probability"]); pr = float(list(dec["codec"].values())[0]["concentration"])
x_noise = noise * rbp / pr / Pc
x_rest = rest / Pc
post = 1 / (1 + x_noise + x_rest)
thr = float(dec["confidence threshold"])
print(name, "N", N, "L", L, "noise", noise, "rbp", rbp, "prior", pr)
print("not strict", (~strict).mean())
ok = strict
print("k hist", np.bincount(k[ok], minlength=6)[:8] / n)
passes_noise = Pc > rbp
print("fails noise filter", (ok & ~passes_noise).mean())
ok2 = ok & passes_noise
print("true rest > 9e-9", (ok2 & (x_rest > 9e-9)).mean(), " true rest > 1e-6:", (ok2 & (x_rest > 1e-6)).mean(), " >5e-4:", (ok2 & (x_rest > 5e-4)).mean())
print("x_noise > 5e-4 (precision)", (ok2 & (x_noise > 5e-4)).mean(), " of which true rest <= 9e-9:", (ok2 & (x_noise > 5e-4) & (x_rest <= 9e-9)).mean())
print("posterior within 2e-8 of threshold", (ok2 & (abs(post - thr) < 2e-8)).mean())
print("low confidence (post <= thr)", (ok2 & (post <= thr)).mean())
for kk in range(0, 5):
    m = ok2 & (k == kk)
    print(" k=%d: share %.4f  x_noise>5e-4: %.4f   true rest>9e-9: %.4f" % (kk, m.mean(), (m & (x_noise > 5e-4)).mean(), (m & (x_rest > 9e-9)).mean()))

# ---- the kernel's own bound
bd = (words[:, None, :] != words[None, :, :]).sum(axis=2) + np.eye(N, dtype=int) * 99
radius = bd.min(axis=1)
cls = [(bd == radius[:, None]).sum(axis=1), (bd == radius[:, None] + 1).sum(axis=1), ((bd >= radius[:, None] + 2) & (bd < 99)).sum(axis=1)]
print("radius histogram", np.bincount(radius)[:14])
MARK = [12, 20, 25, 30, 36]
LOG2 = [-3.8922, -6.6293, -8.3002, -9.9643, -11.9584]
below = [(q < m).sum(axis=1) for m in MARK]
wild = (q < 4).sum(axis=1)
def log2_bound(outside):
    needed = np.maximum(outside - below[0], 0)
    v = 2.0 * wild
    for m in range(4):
        ch = np.minimum(needed, below[m + 1] - below[m]); needed = needed - ch; v = v + LOG2[m] * ch
    return v + LOG2[4] * needed
A = fm.prod(axis=1)
outside = radius[c] - k
xr = (cls[0][c] * 2.0 ** log2_bound(outside) + cls[1][c] * 2.0 ** log2_bound(outside + 1) + cls[2][c] * 2.0 ** log2_bound(outside + 2)) * A / Pc
fin = strict & (k <= 4) & passes_noise & (x_noise + xr <= 3e-4) & (xr <= 9e-9)
print("kernel bound: finished", fin.mean(), " deferred by bound alone", (strict & (k <= 4) & passes_noise & ~((x_noise + xr <= 3e-4) & (xr <= 9e-9))).mean())
m = strict & passes_noise & (xr > 9e-9)
print("of deferred-by-bound: k hist", np.bincount(k[m], minlength=6)[:6] / n, " #below12 hist", np.bincount(below[0][m], minlength=8)[:8] / n, "wild hist", np.bincount(wild[m], minlength=5)[:5] / n)
# better bound: exclude own mismatch positions; own positions may match b (boost 1/rho if rho<1)
# own mismatch positions of each read against c
own = obs != words[c]
def model(marks, exclude_own, wildq=4):
    rho = [10 ** (-m / 10) / (1 - 10 ** (-m / 10)) for m in marks]
    lg = [np.log2(r) for r in rho]
    qq = np.where(own, 99, q) if exclude_own else q
    bel = [(qq < m).sum(axis=1) for m in marks]
    wl = (q < wildq).sum(axis=1)
    def lb(outside):
        needed = np.maximum(outside - bel[0], 0)
        v = 2.0 * wl
        for m in range(len(marks) - 1):
            ch = np.minimum(needed, bel[m + 1] - bel[m]); needed = needed - ch; v = v + lg[m] * ch
        return v + lg[-1] * needed
    xr = (cls[0][c] * 2.0 ** lb(outside) + cls[1][c] * 2.0 ** lb(outside + 1) + cls[2][c] * 2.0 ** lb(outside + 2)) * A / Pc
    ok = strict & (k <= 4) & passes_noise
    fin = ok & (x_noise + xr <= 3e-4) & (xr <= 9e-9)
    bad = ok & (xr * 1.0001 < x_rest)
    if bad.any(): print("   NOT A BOUND for", bad.sum(), "reads; e.g.", np.flatnonzero(bad)[:3])
    model.last = xr
    return fin.mean()
print("marks 5, own included :", model(MARK, False))
print("marks 5, own excluded :", model(MARK, True))
print("marks 8, own excluded :", model([8, 12, 16, 20, 24, 28, 32, 36], True))
print("upper limit (true rest):", (strict & (k <= 4) & passes_noise & (x_rest <= 9e-9)).mean())

a = model(MARK, False); xa = model.last
b = model(MARK, True); xb = model.last
okk = strict & (k <= 4) & passes_noise
print("own excluded made the bound larger for", (okk & (xb > xa * 1.0001)).sum(), "smaller for", (okk & (xb < xa * 0.9999)).sum())
i = np.flatnonzero(okk & (xb > xa * 1.0001))[:1]
for j in i:
    print("read", j, "q", q[j], "own", own[j].astype(int), "k", k[j], "radius", radius[c[j]], "xa", xa[j], "xb", xb[j], "true", x_rest[j])
j = 42025
if n > j:
    print("read", j, "q", q[j], "own", own[j].astype(int), "k", k[j], "c", c[j], "radius", radius[c[j]], "xb", xb[j], "true", x_rest[j], "strict", strict[j])
    o = np.argsort(-P[j])[:3]; print(" top3", o, P[j][o], dist[j][o], "bd", bd[o[0], o[1]])
rem = okk & ~((x_noise + xb <= 3e-4) & (xb <= 9e-9)) & (x_rest <= 9e-9)
print("still deferred though true rest is small:", rem.mean(), " k hist", np.bincount(k[rem], minlength=5)[:5] / n, "wild(not own) hist", np.bincount(((q < 4) & ~own).sum(axis=1)[rem], minlength=5)[:5] / n,
      " below12 not own hist", np.bincount(((q < 12) & ~own).sum(axis=1)[rem], minlength=6)[:6] / n)

# ---- a list tier behind it: c and its K nearest barcodes scored exactly, the others bounded from the K + 1-th nearest's distance on
def listed_model(K):
    order = np.argsort(bd, axis=1, kind="stable")[:, :K]                       # per barcode: its K nearest
    r_out = np.sort(bd, axis=1)[:, K]                                          # distance of the nearest barcode that is NOT listed
    cls_out = [(bd == r_out[:, None]).sum(axis=1) - (np.take_along_axis(bd, order, 1) == r_out[:, None]).sum(axis=1),
               (bd == r_out[:, None] + 1).sum(axis=1), ((bd >= r_out[:, None] + 2) & (bd < 99)).sum(axis=1)]
    listed_sum = np.take_along_axis(P, order[c], 1).sum(axis=1) / Pc
    qq = np.where(own, 99, q)
    rho = [10 ** (-m / 10) / (1 - 10 ** (-m / 10)) for m in MARK]
    bel = [(qq < m).sum(axis=1) for m in MARK]
    def lb(outside):
        needed = np.maximum(outside - bel[0], 0)
        v = 2.0 * wild
        for m in range(4):
            ch = np.minimum(needed, bel[m + 1] - bel[m]); needed = needed - ch; v = v + np.log2(rho[m]) * ch
        return v + np.log2(rho[4]) * needed
    out = r_out[c] - k
    xr = (cls_out[0][c] * 2.0 ** lb(out) + cls_out[1][c] * 2.0 ** lb(out + 1) + cls_out[2][c] * 2.0 ** lb(out + 2)) * A / Pc
    fin = okk & (listed_sum <= 5e-4) & (x_noise + listed_sum + xr <= 5e-4) & (xr <= 9e-9)
    return fin.mean(), np.bincount(r_out)[:12]
for K in (4, 8, 16):
    f, h = listed_model(K)
    print("list tier, K = %2d: finished %.4f   (first unlisted distance histogram %s)" % (K, f, h))
