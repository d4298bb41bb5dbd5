"""Single-fit Cholesky rate against N: device-resident time of a log-likelihood-only fit (assembly + factorisation with one
residual row + finish) on the default route and on the launch-per-step route (CGP_NO_DAG=1), N^3/3 flops over that time.
  python tools/chol_rate.py "2048 4096 8192 16384 32768" [m: prediction points per fit, default 0]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K

M = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for n in (int(v) for v in sys.argv[1].split()):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.15 * n, n))
    y = np.sin(0.05 * x) + 0.1 * rng.standard_normal(n)
    res = []
    for env in ({}, {'CGP_NO_DAG': '1'}):
        os.environ.pop('CGP_NO_DAG', None)
        os.environ.update(env)
        dev = _lib.Device(0)
        dev.set_data(1, x, y)
        xn = np.linspace(0., 0.15 * n, M) if M else None
        ll = dev.fit(K.kernel_params_1d([1., 2.]), K.diag_term(n, np.full(n, 0.1), 0., 0.), xn, None, 0.)[0]
        dev.bench_fit(2, stages=False)
        ms = min(dev.bench_fit(3 if n >= 32768 else 5, stages=False)[0] for _ in range(2))
        res.append((ms, ll))
        dev.close()
    f = n ** 3 / 3. + float(n) * n * (M + 1 if M else 0)
    print('N=%6d  single launch %9.3f ms = %5.1f TFLOP/s (%.3f of 78.6)   launch per step %9.3f ms = %5.1f TFLOP/s   loglik rel diff %.1e'
          % (n, res[0][0], f / res[0][0] * 1e-9, f / res[0][0] * 1e-9 / 78.6, res[1][0], f / res[1][0] * 1e-9,
             abs(res[0][1] - res[1][1]) / abs(res[1][1])), flush=True)
