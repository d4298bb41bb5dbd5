"""Eigen route on NOISE-FREE RBF matrices (numerically rank deficient: the case the route exists for): the low-rank pre-stage
(pivoted Cholesky + r x r Jacobi, csrc/jacobi.hpp) against the full Jacobi sweeps (CGP_JAC_NO_LOWRANK=1).
python tools/eig_lowrank_time.py "<n> ..." [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# modes above `thresh` are kept.  The reference's 1e-15 sits BELOW the rounding noise of any decomposition of these matrices
# (|K| ~ 60: noise ~ 1e-14), where K^+ is not a reproducible quantity; 1e-9 keeps the resolved modes only, and the two routes
# must then agree
thresh = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
for n in (int(v) for v in sys.argv[1].split()):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.1 * n, n))
    d = x[:, None] - x[None, :]
    K = np.exp(-0.5 * d * d / 4.)
    w = np.sin(0.3 * x) + 0.5 * np.cos(0.11 * x)
    y = K @ w                                   # in the range of K: K K^+ y = y is a well-conditioned statement
    res = {}
    for mode in ('lowrank', 'full'):
        if mode == 'full':
            os.environ['CGP_JAC_NO_LOWRANK'] = '1'
        else:
            os.environ.pop('CGP_JAC_NO_LOWRANK', None)
        dev = _lib.Device(0)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            inv, ld, nd = dev.svd_inverse(K, thresh=thresh)
            best = min(best, time.perf_counter() - t0)
        dev.close()
        a = inv @ y
        res[mode] = (best, ld, nd, np.abs(K @ a - y).max() / np.abs(y).max(), inv)
        print('n=%5d %-8s %8.1f ms  logdet %.6e  dropped %4d  max|K K^+ y - y|/max|y| %.2e' % (n, mode, best * 1e3, ld, nd, res[mode][3]), flush=True)
    xs = np.linspace(x[0], x[-1], 50)
    H = np.exp(-0.5 * (xs[:, None] - x[None, :]) ** 2 / 4.)
    p_l, p_f = H @ (res['lowrank'][4] @ y), H @ (res['full'][4] @ y)
    print('        prediction at 50 points: max |low-rank - full| %.2e (max |pred| %.2f); speed-up %.1fx' % (np.abs(p_l - p_f).max(), np.abs(p_f).max(), res['full'][0] / res['lowrank'][0]), flush=True)
