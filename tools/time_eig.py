"""Developer timing of the eigensolver route (svd_method=True) against host LAPACK."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd
from oracle import cosmogp_oracle as orc
dev = cosmogp_amd._lib.default_device()
for n in (128, 256, 512, 1024, 2048, 4096):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., n / 10., n))
    y = np.sin(x / 9.) + 0.1 * rng.standard_normal(n)
    gp = cosmogp_amd.gaussian_process(y, x, kernel='RBF1D', y_err=np.full(n, 0.1))
    hyp = [1., 2.]
    gp.compute_log_likelihood(hyp)            # warm-up (svd_method=True default)
    t0 = time.perf_counter(); gp.compute_log_likelihood(hyp); t1 = time.perf_counter()
    ll = gp.log_likelihood[0]
    gp.compute_log_likelihood(hyp, svd_method=False); gp.compute_log_likelihood(hyp, svd_method=False)
    t2 = time.perf_counter(); gp.compute_log_likelihood(hyp, svd_method=False); t3 = time.perf_counter()
    msg = ''
    if n <= 2048:
        t4 = time.perf_counter(); ref = orc.log_likelihood_gp(y, x, 'RBF1D', hyp, 0., y_err=np.full(n, 0.1), y_mean=0., svd_method=True)[0]; t5 = time.perf_counter()
        msg = '  host dgesdd route %.1f ms  rel diff %.1e' % ((t5 - t4) * 1e3, abs(ll - ref) / abs(ref))
    print('N=%5d  svd route %.1f ms   cholesky route %.2f ms  rel diff %.1e%s' % (
        n, (t1 - t0) * 1e3, (t3 - t2) * 1e3, abs(ll - gp.log_likelihood[0]) / abs(ll), msg), flush=True)
