"""Numerical model of the neighbour list certificate (pamld_near): which share of the C2 reads it finishes.

For every read: c = nearest barcode (Hamming, first on ties), M = mismatch set against c, the LISTED barcodes = c and
the 15 barcodes nearest to c, everything else bounded through radius(c) = distance of the 17th.  Checks the bound
against the true remainder and reports the deferral rate by cause.
"""
import os
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from pheniqs_amd import synthetic

LIST = int(sys.argv[2]) if len(sys.argv) > 2 else 16
TOL = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
job, recipe = synthetic.config_job(os.environ.get("CONFIG", "C2"), "pamld")        # CONFIG=S384: a crowded set
dec = job["sample"]
words, prior = synthetic._codec_arrays(dec)
N, L = words.shape
code, quality, flag, offsets = synthetic.generate_rows(job, n, recipe["seed"])
bam2code = np.full(16, 255, dtype=np.uint8); bam2code[[1, 2, 4, 8]] = [0, 1, 2, 3]
obs = bam2code[code]                                    # 255 = outside ACGT
q = quality.astype(np.int64)
strict = obs != 255
base = 10.0 ** -0.1
fpp = base ** np.arange(128, dtype=np.float64)
tpp = 1.0 - fpp
match_p = np.where(np.arange(128) > 0, tpp, 1.0); mism_p = np.where(np.arange(128) > 0, fpp, 1.0)
uniform = 0.25
fm = np.where(strict, match_p[q & 127], np.where((q & 127) > 0, uniform, 1.0))
fe = np.where(strict, mism_p[q & 127], np.where((q & 127) > 0, uniform, 1.0))
rho = fe / fm
noise = float(dec["noise"]); conf_thr = float(dec["confidence threshold"])
pri = np.array([float(b["concentration"]) for b in dec["codec"].values()])
random_p = 0.25 ** L
adjusted_noise = noise * random_p
print("priors", pri[:3], "noise", noise, "threshold", conf_thr, "N", N, "L", L)

mis = (obs[:, None, :] != words[None, :, :])            # n x N x L  (outside mismatches everything)
P = np.where(mis, fe[:, None, :], fm[:, None, :]).prod(axis=2)         # n x N conditional
dist = mis.sum(axis=2)
c = dist.argmin(axis=1)
k = dist[np.arange(n), c]
# per barcode: neighbour order and radius
bd = (words[:, None, :] != words[None, :, :]).sum(axis=2)
order = np.argsort(bd, axis=1, kind="stable")          # order[c][0] == c
listed = order[:, :LIST]
radius = bd[np.arange(N), order[:, LIST]]              # distance of the first barcode NOT listed
print("radius histogram", np.bincount(radius))
Pl = np.take_along_axis(P, listed[c], axis=1)          # n x LIST
Pc = Pl[:, 0]
others_listed = Pl[:, 1:].sum(axis=1)
inlist = np.zeros((N, N), dtype=bool); inlist[np.arange(N)[:, None], listed] = True
true_rest = np.where(inlist[c], 0.0, P).sum(axis=1)
# bound on each not listed barcode: A * prod_{j in M} max(rho_j, 1) * top-(radius - k) rho product over all positions (wild ones free)
A = fm.prod(axis=1)
Mmask = np.take_along_axis(mis, c[:, None, None], axis=1)[:, 0, :]
boost = np.where(Mmask, np.maximum(rho, 1.0), 1.0).prod(axis=1)
need = np.maximum(radius[c] - k, 0)
rs = -np.sort(-(np.where(Mmask, 0.0, rho) if os.environ.get('EXCLUDE_M', '1') == '1' else rho), axis=1)
top = np.ones(n)
for i in range(L):
    take = (i < need) | (rs[:, i] > 1.0)
    top *= np.where(take, np.where(rs[:, i] > 0, rs[:, i], 1.0 if True else 0), 1.0)
# a need larger than the positions outside M cannot happen (radius <= L)
U1 = A * boost * top
top2 = np.ones(n)
for i in range(L):
    take = (i < need + 1) | (rs[:, i] > 1.0)
    top2 *= np.where(take, np.where(rs[:, i] > 0, rs[:, i], 1.0), 1.0)
unlisted_at_radius = np.array([(bd[b, order[b, LIST:]] == radius[b]).sum() for b in range(N)])
import os
if os.environ.get("CLASS_BOUND", "1") == "1":
    U = A * boost * (unlisted_at_radius[c] * top + (N - LIST - unlisted_at_radius[c]) * top2)
elif os.environ.get("CLASS_BOUND") == "pos":
    # k = 1: only the unlisted barcodes at the radius whose difference from c covers the read's mismatch position can
    # come one mismatch closer
    diff = (words[:, None, :] != words[None, :, :])                   # N x N x L
    unl = ~inlist                                                       # N x N
    at_r = unl & (bd == radius[:, None])
    m_pos = (at_r[:, :, None] & diff).sum(axis=1)                       # N x L: unlisted at the radius differing at position j
    j0 = Mmask.argmax(axis=1)
    mj = m_pos[c, j0]
    k1 = (k == 1)
    # top (need) for k=1 is r-1; top2 is r
    U_k1 = A * boost * (mj * top + (unlisted_at_radius[c] - mj) * top2 + (N - LIST - unlisted_at_radius[c]) * top2)
    U_k0 = A * boost * (unlisted_at_radius[c] * top + (N - LIST - unlisted_at_radius[c]) * top2)
    U = np.where(k1, U_k1, U_k0)
elif os.environ.get("CLASS_BOUND") == "3":
    top3 = np.ones(n)
    for i in range(L):
        take = (i < need + 2) | (rs[:, i] > 1.0)
        top3 *= np.where(take, np.where(rs[:, i] > 0, rs[:, i], 1.0), 1.0)
    at_next = np.array([(bd[b, order[b, LIST:]] == radius[b] + 1).sum() for b in range(N)])
    U = A * boost * (unlisted_at_radius[c] * top + at_next[c] * top2 + (N - LIST - unlisted_at_radius[c] - at_next[c]) * top3)
elif os.environ.get("CLASS_BOUND") == "oracle":
    U = true_rest.copy()
else:
    U = (N - LIST) * U1
bad = np.nonzero(true_rest > U * (1 + 1e-9))[0]
if bad.size:
    i = bad[0]
    print("violations", bad.size, "first", i, "c", c[i], "k", k[i], "radius", radius[c[i]], "need", need[i])
    print(" obs", obs[i], "q", q[i], "word c", words[c[i]])
    print(" rho", rho[i], "M", Mmask[i], "A", A[i], "boost", boost[i], "top", top[i], "U1", U1[i], "rest", true_rest[i])
    notl = np.setdiff1d(np.arange(N), listed[c[i]])
    j = notl[np.argmax(P[i, notl])]
    print(" largest not listed", j, words[j], "P", P[i, j], "dist to c", bd[c[i], j], "dist to read", dist[i, j])
    raise SystemExit(1)
print("bound / true remainder: median %.3g" % np.median(U / np.maximum(true_rest, 1e-300)))
pc = Pc * pri[c]
total_lo = pc + others_listed * pri[c] + adjusted_noise
total_hi = total_lo + U * pri[c]
conf_hi = pc / total_lo; conf_lo = pc / total_hi
width = conf_hi - conf_lo
exact_conf = (P * pri).max(axis=1) / ((P * pri).sum(axis=1) + adjusted_noise)
winner_true = (P * pri).argmax(axis=1)
noise_pass = Pc > random_p
dominance = (others_listed + U) <= 5e-4 * Pc
argmax_sure = (others_listed + U) <= 0.9 * Pc
allowed = need >= int(os.environ.get('MIN_NEED', '0'))
XMAX = float(os.environ.get('XMAX', '1e9'))
x_hi = (others_listed + U) / Pc + adjusted_noise / (pri[c] * Pc)
allowed &= x_hi <= XMAX
print('share with x <= XMAX: %.3f %%' % (100 * (x_hi <= XMAX).mean()))
finish = allowed & np.where(noise_pass, dominance & (width <= TOL) & (np.abs(conf_hi - conf_thr) > 1e-9 + width),
                  argmax_sure)
assert (winner_true[finish] == c[finish]).all()
assert (np.abs(conf_hi - exact_conf)[finish & noise_pass] <= TOL * 1.01).all()
print("reads %d: finished %.3f %%, deferred %.3f %%" % (n, 100 * finish.mean(), 100 * (1 - finish.mean())))
for kk in range(0, 5):
    sel = k == kk
    if sel.any():
        print("  k=%d: share %.3f %%, deferred %.3f %% of them (%.3f %% of all)" % (kk, 100 * sel.mean(), 100 * (1 - finish[sel].mean()), 100 * (sel & ~finish).mean()))
d = ~finish
print("  deferred: noise_pass & !dominance %.3f %%, width %.3f %%, threshold %.3f %%, !noise_pass & !argmax %.3f %%" % (
    100 * (d & noise_pass & ~dominance).mean(), 100 * (d & noise_pass & dominance & (width > TOL)).mean(),
    100 * (d & noise_pass & dominance & (width <= TOL)).mean(), 100 * (d & ~noise_pass).mean()))
# what the current product form kernel defers (others <= 5e-4 winner with the exact sum)
others_all = (P * pri).sum(axis=1) - (P * pri).max(axis=1)
print("for comparison, exact dominance fails on %.3f %%" % (100 * ((others_all > 5e-4 * (P * pri).max(axis=1)) & ((P.max(axis=1)) > random_p)).mean()))
