"""Developer timing: likelihood vs likelihood+gradient evaluations, explicit inverse (operator level)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd
from cosmogp_amd import _lib

for n, dim in ((2048, 1), (4096, 1), (8192, 2)):
    rng = np.random.default_rng(n)
    if dim == 1:
        x = np.sort(rng.uniform(0., n / 10., n)); hyp = [1., 2.]
        y = np.sin(x / 9.) + 0.1 * rng.standard_normal(n)
        gp = cosmogp_amd.gaussian_process(y, x, kernel='RBF1D', y_err=np.full(n, 0.1))
    else:
        x = rng.uniform(0., 2048., (n, 2)); hyp = [1., 110., 110., 0.]
        y = np.sin(x[:, 0] / 200.) + 0.1 * rng.standard_normal(n)
        gp = cosmogp_amd.gaussian_process(y, x, kernel='RBF2D', y_err=np.full(n, 0.1))
    gp.compute_log_likelihood(hyp, svd_method=False); gp.compute_log_likelihood_gradient(hyp)
    t0 = time.perf_counter()
    for _ in range(5): gp.compute_log_likelihood(hyp, svd_method=False)
    t1 = time.perf_counter()
    for _ in range(5): gp.compute_log_likelihood_gradient(hyp)
    t2 = time.perf_counter()
    K = gp.kernel(gp.Time[0], hyp, y_err=gp.y_err[0]) if n <= 4096 else None
    msg = ''
    if K is not None:
        cosmogp_amd.cholesky_inverse(K)
        t3 = time.perf_counter(); inv = cosmogp_amd.cholesky_inverse(K); t4 = time.perf_counter()
        msg = ' cholesky_inverse (host in/out) %.1f ms, |K inv - I| %.2e' % ((t4 - t3) * 1e3, np.abs(K @ inv - np.eye(n)).max())
    print('N=%d dim=%d  loglik %.2f ms   loglik+gradient %.2f ms%s' % (n, dim, (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3, msg), flush=True)
