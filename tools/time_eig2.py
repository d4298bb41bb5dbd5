import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd
n = int(sys.argv[1])
rng = np.random.default_rng(n)
x = np.sort(rng.uniform(0., n / 10., n))
y = np.sin(x / 9.) + 0.1 * rng.standard_normal(n)
gp = cosmogp_amd.gaussian_process(y, x, kernel='RBF1D', y_err=np.full(n, 0.1))
gp.compute_log_likelihood([1., 2.])
t0 = time.perf_counter(); gp.compute_log_likelihood([1., 2.]); t1 = time.perf_counter()
ll = gp.log_likelihood[0]
gp.compute_log_likelihood([1., 2.], svd_method=False)
print('N=%d svd route %.1f ms rel diff vs cholesky %.1e' % (n, (t1 - t0) * 1e3, abs(ll - gp.log_likelihood[0]) / abs(ll)))
