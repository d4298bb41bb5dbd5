"""Developer timing of the operator-level calls (N x N matrices cross PCIe)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd as cg
rng = np.random.default_rng(0)
for n in (1024, 4096, 8192):
    x = np.sort(rng.uniform(0., n / 10., n)); e = np.full(n, 0.1)
    K = cg.rbf_kernel_1d(x, [1., 2.], y_err=e)
    t0 = time.perf_counter(); K = cg.rbf_kernel_1d(x, [1., 2.], y_err=e); t1 = time.perf_counter()
    msg = 'N=%d rbf_kernel_1d -> host %.1f ms (%.1f GB/s)' % (n, (t1 - t0) * 1e3, n * n * 8 / (t1 - t0) / 1e9)
    if n <= 4096:
        inv = cg.cholesky_inverse(K)
        t2 = time.perf_counter(); inv = cg.cholesky_inverse(K); t3 = time.perf_counter()
        msg += '   cholesky_inverse host->host %.1f ms' % ((t3 - t2) * 1e3)
    print(msg, flush=True)
