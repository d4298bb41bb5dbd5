"""Developer check: one log-likelihood at N = 65536 (34 GB matrix) against the sum of its two far-apart halves."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd as cg
rng = np.random.default_rng(78)
n = 32768
xa = np.sort(rng.uniform(0., 6000., n)); xb = np.sort(rng.uniform(1e6, 1e6 + 6000., n))
f = lambda x: np.sin(x / 7.) + 0.3 * np.cos(x / 1.3)
ya = f(xa) + 0.1 * rng.standard_normal(n); yb = f(xb - 1e6) + 0.1 * rng.standard_normal(n)
e = np.full(n, 0.1); hyp = [1.0, 2.0]
lls = []
for x, y in ((xa, ya), (xb, yb), (np.concatenate([xa, xb]), np.concatenate([ya, yb]))):
    gp = cg.gaussian_process(y, x, kernel='RBF1D', y_err=np.full(len(x), 0.1))
    t0 = time.perf_counter(); gp.compute_log_likelihood(hyp, svd_method=False); t1 = time.perf_counter()
    lls.append(gp.log_likelihood[0]); print('N=%d loglik %.9f  %.2f s  (%.1f TFLOP/s Cholesky)' % (len(x), lls[-1], t1 - t0, len(x) ** 3 / 3. / (t1 - t0) / 1e12), flush=True)
print('sum of halves %.9f  whole %.9f  rel diff %.2e' % (lls[0] + lls[1], lls[2], abs(lls[2] - lls[0] - lls[1]) / abs(lls[2])))
