"""Eigen route on a FULL-RANK matrix (RBF + noise): certified Cholesky shortcut against the Jacobi sweeps.
python tools/eig_fullrank_time.py "<sizes>"   (fused entry point cgp_fit_svd: no N x N matrix crosses PCIe)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K

sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else '1024 2048 4096').split()]
for n in sizes:
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 0.1 * n, n))
    y = np.sin(0.05 * x) + 0.1 * rng.standard_normal(n)
    kp = K.kernel_params_1d([1., 2.], 0.)
    diag = K.diag_term(n, np.full(n, 0.1), 0., 0.)
    xn = np.linspace(0., 0.1 * n, 64)
    res = {}
    for mode in ('shortcut', 'sweeps'):
        if mode == 'sweeps':
            os.environ['CGP_EIG_NO_CHOL'] = '1'
        else:
            os.environ.pop('CGP_EIG_NO_CHOL', None)
        dev = _lib.Device(0)
        dev.set_data(1, x, y)
        out = dev.fit_svd(kp, diag, xn, None, 0., 1e-15)
        t0 = time.perf_counter()
        reps = 3 if mode == 'shortcut' else 1
        for _ in range(reps):
            out = dev.fit_svd(kp, diag, xn, None, 0., 1e-15)
        dt = (time.perf_counter() - t0) / reps
        res[mode] = (dt, out, dev.eig_info())
        dev.close()
    (t2, o2, r2), (t0_, o0, r0) = res['shortcut'], res['sweeps']
    dmu = float(np.max(np.abs(np.asarray(o2[3]) - np.asarray(o0[3]))))
    print('n %5d: shortcut %8.2f ms (route %d)   sweeps %9.2f ms (route %d, %d sweeps)   loglik %.12g / %.12g   max |mu diff| %.2e'
          % (n, t2 * 1e3, r2[0], t0_ * 1e3, r0[0], r0[2], o2[0], o0[0], dmu), flush=True)
