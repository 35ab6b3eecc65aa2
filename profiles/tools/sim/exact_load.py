"""What reaches the exact tier behind C2's streaming kernel, by reason (model: fp64 numpy of the same decisions)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pheniqs_amd import synthetic, capi
job, recipe = synthetic.config_job("C2", "pamld")
dec = job["sample"]; words, prior = synthetic._codec_arrays(dec); N, L = words.shape
n = 400000
code, quality, flag, offsets = synthetic.generate_rows(job, n, 11)
bam2code = np.full(16, 255, dtype=np.uint8); bam2code[[1, 2, 4, 8]] = [0, 1, 2, 3]
obs = bam2code[code]; q = quality.astype(np.int64)
fpp = (10.0 ** -0.1) ** np.arange(128.0); tpp = 1 - fpp
outside = obs == 255
fm = np.where(outside, 0.25, np.where(q > 0, tpp[q], 1.0)); fe = np.where(outside, 0.25, np.where(q > 0, fpp[q], 1.0))
mis = (obs[:, None, :] != words[None, :, :]) | outside[:, None, :]
p = np.where(mis, fe[:, None, :], fm[:, None, :]).prod(axis=2)
pri = float(dec["codec"][next(iter(dec["codec"]))]["concentration"]); noise = float(dec["noise"]); rbp = 0.25 ** L
order = np.argsort(-p, axis=1)
best = p[np.arange(n), order[:, 0]]; second = p[np.arange(n), order[:, 1]]; third = p[np.arange(n), order[:, 2]]
others = p.sum(axis=1) - best
is_noise = best <= rbp
dom = 9e-9 / (2 * 4.3e-7 + 1e-6)
tie32 = second > 0.9999 * best
print("reads %d, noise reads %.3f %%" % (n, 100 * is_noise.mean()))
print("noise with a tie in single precision: %.3f %% of all" % (100 * (is_noise & tie32).mean()))
cls = ~is_noise
notdom = cls & (others > dom * best)
print("classified, others > dominance(%.2e) x best: %.3f %%" % (dom, 100 * notdom.mean()))
rest3 = others - second
print("  of those, everything but the runner up <= dominance x best: %.3f %% (a runner up in fp64 would finish them)" % (100 * (notdom & (rest3 <= dom * best)).mean()))
print("  of those, everything but runner up and third <= dominance x best: %.3f %%" % (100 * (notdom & (rest3 - third <= dom * best)).mean()))
conf = best * pri / (p.sum(axis=1) * pri + noise * rbp)
print("within 2e-8 of the threshold: %.4f %%" % (100 * (cls & (np.abs(conf - float(dec["confidence threshold"])) <= 2e-8)).mean()))
