"""Randomised differential test (fixed seeds): the public API on random shapes / kernels / options against the
oracle.  Complements the fixed-case parity tests; sizes stay where the dense oracle takes milliseconds."""
import numpy as np
import pytest

from oracle import cosmogp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cg():
    import cosmogp_amd
    cosmogp_amd._lib.default_device()
    return cosmogp_amd


def _object(rng, dim, n):
    if dim == 1:
        x = np.sort(rng.uniform(0., 0.3 * n + 5., n))
        f = np.sin(x / 3.)
    else:
        x = rng.uniform(0., 2. * np.sqrt(n) + 5., (n, 2))
        f = np.sin(x[:, 0] / 4.) * np.cos(x[:, 1] / 5.)
    e = rng.uniform(0.05, 0.2, n)
    return x, f + e * rng.standard_normal(n), e


@pytest.mark.parametrize('seed', range(24))
def test_random_config(cg, seed):
    rng = np.random.default_rng(1000 + seed)
    dim = 1 + seed % 2
    kernel = 'RBF1D' if dim == 1 else 'RBF2D'
    nobj = int(rng.choice([1, 1, 2, 5]))
    ns = [int(rng.integers(1, 420)) for _ in range(nobj)]
    objs = [_object(rng, dim, n) for n in ns]
    xs, ys, es = ([o[k] for o in objs] for k in range(3))
    hyp = [float(rng.uniform(0.5, 1.5)), float(rng.uniform(1.5, 4.))] if dim == 1 else \
          [float(rng.uniform(0.5, 1.5)), float(rng.uniform(3., 6.)), float(rng.uniform(3., 6.)), float(rng.uniform(-4., 4.))]
    nugget = float(rng.choice([0., 0.03, 0.2]))
    own_grid = bool(rng.integers(0, 2))
    m = int(rng.integers(1, 200))
    if own_grid:
        grid = None
    else:
        grid = np.sort(rng.uniform(0., 30., m)) if dim == 1 else rng.uniform(0., 30., (m, 2))
    svd = seed % 6 == 5                      # a few through the reference's default route
    gpm = __import__('sys').modules['cosmogp_amd.gaussian_process']
    gp = (cg.gaussian_process_nobject(ys, xs, kernel=kernel, y_err=es) if nobj > 1 else
          cg.gaussian_process(ys[0], xs[0], kernel=kernel, y_err=es[0]))
    gp.hyperparameters = hyp
    gp.nugget = nugget
    old = gpm.SVD_VIA_CHOLESKY_WHEN_PD
    try:
        if svd:
            gpm.SVD_VIA_CHOLESKY_WHEN_PD = False
        gp.get_prediction(new_binning=grid, svd_method=svd)
        gp.compute_log_likelihood(hyp, svd_method=svd)
    finally:
        gpm.SVD_VIA_CHOLESKY_WHEN_PD = old
    want = orc.predict(ys, xs, kernel, hyp, nugget, es, new_binning=grid, svd_method=svd)
    tol = 1e-8 if svd else 1e-9
    for i in range(nobj):
        np.testing.assert_allclose(gp.Prediction[i], want['prediction'][i], rtol=0, atol=tol)
        np.testing.assert_allclose(gp.variance[i], np.diag(want['covariance'][i]), rtol=0, atol=tol)
    k = int(rng.integers(0, nobj))
    np.testing.assert_allclose(gp.covariance_matrix[k], want['covariance'][k], rtol=0, atol=tol)
    ll = orc.total_log_likelihood(ys, xs, kernel, hyp, nugget, es, np.zeros(nobj), svd_method=svd)
    np.testing.assert_allclose(gp.log_likelihood, ll, rtol=1e-10)
    if seed % 3 == 0 and not svd:            # analytic gradient, nugget column included
        gp.fit_nugget = True
        g = gp.compute_log_likelihood_gradient(hyp + [nugget])
        ref = sum(orc.log_likelihood_gradient(ys[i], xs[i], kernel, hyp, nugget, y_err=es[i], y_mean=0.) for i in range(nobj))
        np.testing.assert_allclose(g, ref, rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize('seed', range(8))
def test_random_call_sequences_on_one_handle(cg, seed):
    """State-machine fuzz: three GP objects share the default handle and take turns with every kind of call
    (likelihood, prediction on shared / own grids, lazy matrices, gradient, sweep, eigensolver route, operator-level
    inverse); every result is checked against the oracle, so state leaking from one call into the next shows up."""
    from cosmogp_amd import multigpu
    gpm = __import__('sys').modules['cosmogp_amd.gaussian_process']
    rng = np.random.default_rng(500 + seed)
    o1 = _object(rng, 1, int(rng.integers(40, 300)))
    o2 = _object(rng, 2, int(rng.integers(40, 300)))
    many = [_object(rng, 1, int(rng.integers(5, 330))) for _ in range(4)]
    gps = [
        ('RBF1D', [o1[1]], [o1[0]], [o1[2]], cg.gaussian_process(o1[1], o1[0], kernel='RBF1D', y_err=o1[2])),
        ('RBF2D', [o2[1]], [o2[0]], [o2[2]], cg.gaussian_process(o2[1], o2[0], kernel='RBF2D', y_err=o2[2])),
        ('RBF1D', [m[1] for m in many], [m[0] for m in many], [m[2] for m in many],
         cg.gaussian_process_nobject([m[1] for m in many], [m[0] for m in many], kernel='RBF1D', y_err=[m[2] for m in many])),
    ]
    for step in range(30):
        kernel, ys, xs, es, gp = gps[int(rng.integers(0, 3))]
        nobj = len(ys)
        hyp = [float(rng.uniform(0.6, 1.4)), float(rng.uniform(1.5, 4.))] if kernel == 'RBF1D' else \
              [float(rng.uniform(0.6, 1.4)), float(rng.uniform(3., 6.)), float(rng.uniform(3., 6.)), float(rng.uniform(-3., 3.))]
        nugget = float(rng.choice([0., 0.05]))
        gp.hyperparameters, gp.nugget, gp.fit_nugget = hyp, nugget, False
        op = int(rng.integers(0, 8))
        if op == 0:
            gp.compute_log_likelihood(hyp, svd_method=False)
            ref = orc.total_log_likelihood(ys, xs, kernel, hyp, nugget, es, np.zeros(nobj), svd_method=False)
            np.testing.assert_allclose(gp.log_likelihood, ref, rtol=1e-10)
        elif op in (1, 2):
            m = int(rng.integers(1, 150))
            grid = None if op == 2 else (np.sort(rng.uniform(0., 30., m)) if kernel == 'RBF1D' else rng.uniform(0., 30., (m, 2)))
            gp.get_prediction(new_binning=grid, svd_method=False)
            want = orc.predict(ys, xs, kernel, hyp, nugget, es, new_binning=grid, svd_method=False)
            k = int(rng.integers(0, nobj))
            np.testing.assert_allclose(gp.Prediction[k], want['prediction'][k], rtol=0, atol=1e-9)
            np.testing.assert_allclose(gp.covariance_matrix[k], want['covariance'][k], rtol=0, atol=1e-9)
            if step % 3 == 0:
                np.testing.assert_allclose(gp.inv_kernel_matrix[k], want['inv_kernel_matrix'][k], rtol=0,
                                           atol=1e-8 * np.abs(want['inv_kernel_matrix'][k]).max())
        elif op == 3:
            g = gp.compute_log_likelihood_gradient(hyp)
            ref = sum(orc.log_likelihood_gradient(ys[i], xs[i], kernel, hyp, nugget, y_err=es[i], y_mean=0.) for i in range(nobj))
            np.testing.assert_allclose(g, ref[:len(hyp)], rtol=1e-7, atol=1e-7)
        elif op == 4 and nobj == 1:
            table = np.array([[h * f for h in hyp] for f in (0.9, 1.0, 1.1)])
            if kernel == 'RBF2D':
                table[:, 3] = hyp[3]
            got = multigpu.sweep_log_likelihood(gp, table)
            ref = [orc.total_log_likelihood(ys, xs, kernel, list(row), nugget, es, np.zeros(1), svd_method=False)[0] for row in table]
            np.testing.assert_allclose(got, ref, rtol=1e-10)
        elif op == 5:
            old = gpm.SVD_VIA_CHOLESKY_WHEN_PD
            try:
                gpm.SVD_VIA_CHOLESKY_WHEN_PD = False
                gp.compute_log_likelihood(hyp, svd_method=True)
            finally:
                gpm.SVD_VIA_CHOLESKY_WHEN_PD = old
            ref = orc.total_log_likelihood(ys, xs, kernel, hyp, nugget, es, np.zeros(nobj), svd_method=True)
            np.testing.assert_allclose(gp.log_likelihood, ref, rtol=1e-9)
        elif op == 6:
            k = int(rng.integers(0, nobj))
            K = gp.kernel(xs[k], hyp, nugget=nugget, y_err=es[k])
            inv, ld = cg.cholesky_inverse(K, return_logdet=True)
            ref, rld = orc.cholesky_inverse(K, return_logdet=True)
            np.testing.assert_allclose(inv, ref, rtol=0, atol=1e-8 * np.abs(ref).max())
            assert abs(ld - rld) <= 1e-10 * max(1., abs(rld))
        else:
            gp.get_prediction(new_binning=None, COV=False)          # reference defaults (svd_method=True -> 'auto')
            want = orc.predict(ys, xs, kernel, hyp, nugget, es, new_binning=None, svd_method=True, cov=False)
            k = int(rng.integers(0, nobj))
            np.testing.assert_allclose(gp.Prediction[k], want['prediction'][k], rtol=0, atol=1e-8)
