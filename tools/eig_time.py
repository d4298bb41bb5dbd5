"""Time of the eigensolver route (cgp_svd_inverse) on an RBF + noise matrix: python tools/eig_time.py "<n> ..." [reps]
(CGP_DEBUG=1 prints the sweeps; under rocprofv3 --kernel-trace --stats the per-kernel split)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib

dev = _lib.Device(0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for n in (int(v) for v in sys.argv[1].split()):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.1 * n, n))
    d = x[:, None] - x[None, :]
    K = np.exp(-0.5 * d * d / 4.) + 1e-2 * np.eye(n)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        inv, ld, nd = dev.svd_inverse(K)
        best = min(best, time.perf_counter() - t0)
    sign, ldr = np.linalg.slogdet(K)
    err = np.abs(K @ inv - np.eye(n)).max() if n <= 4096 else float('nan')
    print('n=%5d  %.1f ms  logdet rel err %.2e  max|K K^+ - I| %.2e  dropped %d' % (n, best * 1e3, abs(ld - ldr) / abs(ldr), err, nd), flush=True)
