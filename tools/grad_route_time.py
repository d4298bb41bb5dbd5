import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
for n in (int(v) for v in sys.argv[1].split()):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.15 * n, n)); y = np.sin(0.05 * x) + 0.1 * rng.standard_normal(n)
    out = []
    for env in ({}, {'CGP_NO_DAG': '1'}):
        os.environ.pop('CGP_NO_DAG', None); os.environ.update(env)
        dev = _lib.Device(0); dev.set_data(1, x, y)
        kp = K.kernel_params_1d([1., 2.]); diag = K.diag_term(n, np.full(n, 0.1), 0., 0.)
        r = dev.fit_grad(kp, diag)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = dev.fit_grad(kp, None); best = min(best, time.perf_counter() - t0)
        out.append((best, r)); dev.close()
    a, b = out[0][1], out[1][1]
    print('n=%5d grad: single launch %8.3f ms  launch per step %8.3f ms   loglik rel diff %.1e  sums rel diff %.1e'
          % (n, out[0][0] * 1e3, out[1][0] * 1e3, abs(a[0] - b[0]) / abs(b[0]), np.max(np.abs(np.asarray(a[-1]) - np.asarray(b[-1])) / (np.abs(np.asarray(b[-1])) + 1e-300))), flush=True)
