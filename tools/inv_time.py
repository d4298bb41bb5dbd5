import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from cosmogp_amd import _lib
dev = _lib.Device(0)
for n in (1024, 2048, 4096, 8192):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.1 * n, n)); d = x[:, None] - x[None, :]
    K = np.exp(-0.5 * d * d / 4.) + 1e-2 * np.eye(n)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); inv, ld = dev.cholesky_inverse(K)[:2]; best = min(best, time.perf_counter() - t0)
    print('n=%5d cholesky_inverse %.2f ms (incl. %d MB up + down)  |K inv - I| %.1e' % (n, best * 1e3, 2 * n * n * 8 >> 20, np.abs(K @ inv - np.eye(n)).max()), flush=True)
