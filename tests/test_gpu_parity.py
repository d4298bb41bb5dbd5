"""Parity tests proper (run with -m gpu on an MI355X): every call goes through the C ABI of
libcosmogp_hip.so and is compared with (i) golden vectors produced by the real reference,
(ii) the CPU oracle on the same seeded inputs, (iii) size-independent properties at
BASELINE.json's full sizes.

Tolerances (fp64; SURVEY.md 8c, frozen here):
  K / H elementwise        <= 4 ulp   (device exp vs libm; everything else is bit-identical)
  logdet, log-likelihood   rel 1e-11
  prediction               1e-9 * max|y| ; predictive variance 1e-9 * sigma^2   (cond(K) <= 1e6)
  explicit inverse         rel 1e-8 (of max |K^-1|)
"""
import numpy as np
import pytest

from conftest import assert_within_bounds, load_golden, oracle_fit_with_bounds, ulp_diff
from oracle import cosmogp_oracle as orc

pytestmark = pytest.mark.gpu

ULP = 4


@pytest.fixture(scope='module')
def cg():
    import cosmogp_amd
    cosmogp_amd.default_device().selftest_mfma()
    return cosmogp_amd


def objs(g, keys='xye', prefix=''):
    n = int(g['nobj'])
    return ([g['%s%s_%d' % (prefix, k, i)] for i in range(n)] for k in keys)


# ----------------------------------------------------------------- a1 / a2 kernels
@pytest.mark.parametrize('n', [8, 64, 256])
def test_rbf_kernel_1d_golden(cg, n):
    g = load_golden('k1d_n%d' % n)
    ng, fl = float(g['nugget']), float(g['floor'])
    assert ulp_diff(cg.rbf_kernel_1d(g['x'], g['hyp']), g['K_plain']) <= ULP
    assert ulp_diff(cg.rbf_kernel_1d(g['x'], g['hyp'], nugget=ng, floor=fl, y_err=g['yerr']), g['K_noise']) <= ULP
    H = cg.rbf_kernel_1d(g['x'], g['hyp'], new_x=g['xnew'])
    assert H.shape == g['H'].shape and H.flags['C_CONTIGUOUS'] and ulp_diff(H, g['H']) <= ULP
    assert ulp_diff(cg.rbf_kernel_1d(g['xnew'], g['hyp'], nugget=ng), g['Kss']) <= ULP


@pytest.mark.parametrize('n', [8, 64, 200])
def test_rbf_kernel_2d_golden(cg, n):
    g = load_golden('k2d_n%d' % n)
    ng, fl = float(g['nugget']), float(g['floor'])
    assert ulp_diff(cg.rbf_kernel_2d(g['x'], g['hyp']), g['K_plain']) <= ULP
    assert ulp_diff(cg.rbf_kernel_2d(g['x'], g['hyp'], nugget=ng, floor=fl, y_err=g['yerr']), g['K_noise']) <= ULP
    assert ulp_diff(cg.rbf_kernel_2d(g['x'], g['hyp'], new_x=g['xnew']), g['H']) <= ULP
    assert ulp_diff(cg.rbf_kernel_2d(g['xnew'], g['hyp'], nugget=ng), g['Kss']) <= ULP


def test_rbf_kernel_edge_shapes(cg):
    rng = np.random.default_rng(5)
    for n, m in ((1, 1), (2, 3), (127, 1), (128, 129), (129, 128), (1000, 7)):
        x = rng.uniform(0, 30, n)
        xn = rng.uniform(0, 30, m)
        e = rng.uniform(0.1, 0.2, n)
        hyp = [0.9, 2.5]
        assert ulp_diff(cg.rbf_kernel_1d(x, hyp, y_err=e), orc.rbf_kernel_1d(x, hyp, y_err=e)) <= ULP
        assert ulp_diff(cg.rbf_kernel_1d(x, hyp, new_x=xn), orc.rbf_kernel_1d(x, hyp, new_x=xn)) <= ULP
        X = rng.uniform(0, 30, (n, 2))
        Xn = rng.uniform(0, 30, (m, 2))
        h2 = [0.9, 2.5, 3.5, 0.8]
        assert ulp_diff(cg.rbf_kernel_2d(X, h2, y_err=e), orc.rbf_kernel_2d(X, h2, y_err=e)) <= ULP
        assert ulp_diff(cg.rbf_kernel_2d(X, h2, new_x=Xn), orc.rbf_kernel_2d(X, h2, new_x=Xn)) <= ULP
    # identical points / far-apart points (exp underflow) behave like NumPy
    x = np.array([0., 0., 1e3, 1e3 + 1e-9])
    assert ulp_diff(cg.rbf_kernel_1d(x, [1., 1.]), orc.rbf_kernel_1d(x, [1., 1.])) <= ULP


def test_sigma2_switch_2d(cg):
    from cosmogp_amd import kernel as K
    g = load_golden('k2d_n64')
    try:
        K.SIGMA2_IN_TRAIN_2D = True
        got = cg.rbf_kernel_2d(g['x'], g['hyp'], y_err=g['yerr'])
    finally:
        K.SIGMA2_IN_TRAIN_2D = False
    assert ulp_diff(got, orc.rbf_kernel_2d(g['x'], g['hyp'], y_err=g['yerr'], sigma2_in_train=True)) <= ULP


# ----------------------------------------------------------------- a4 inverse
def test_cholesky_inverse_golden(cg):
    g = load_golden('inverse_n96')
    K0 = g['K'].copy()
    inv, ld = cg.cholesky_inverse(g['K'], return_logdet=True)
    assert np.array_equal(g['K'], K0)                          # input untouched
    assert inv.shape == K0.shape and inv.flags['C_CONTIGUOUS']
    np.testing.assert_allclose(inv, g['chol_inv'], rtol=0, atol=1e-8 * np.abs(g['chol_inv']).max())
    assert abs(ld - float(g['chol_logdet'])) <= 1e-11 * abs(ld)
    assert np.allclose(inv, inv.T, rtol=0, atol=1e-9 * np.abs(inv).max())
    inv2 = cg.cholesky_inverse(g['K'])
    assert np.array_equal(inv, inv2)                           # deterministic
    with pytest.raises(np.linalg.LinAlgError):                 # reference raises too (golden flag)
        cg.cholesky_inverse(g['K_singular'])
    assert int(g['chol_raises_on_singular']) == 1


@pytest.mark.parametrize('n', [1, 2, 17, 128, 129, 300, 1000])
def test_cholesky_inverse_sizes(cg, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    K = A @ A.T + n * np.eye(n)
    inv, ld = cg.cholesky_inverse(K, return_logdet=True)
    ref, rld = orc.cholesky_inverse(K, return_logdet=True)
    np.testing.assert_allclose(inv, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    assert abs(ld - rld) <= 1e-11 * max(1., abs(rld))


def test_cholesky_not_pd_reports_like_lapack(cg):
    K = np.eye(200)
    K[150, 150] = -1.
    with pytest.raises(np.linalg.LinAlgError) as ei:
        cg.cholesky_inverse(K)
    assert '151' in str(ei.value)


# ----------------------------------------------------------------- a6 / a7 log-likelihood
def test_loglik_1d_golden(cg):
    g = load_golden('loglik_1d')
    xs, ys, es = objs(g)
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    gp.nugget = float(g['nugget'])
    gp.compute_log_likelihood(g['hyp'], svd_method=False)          # batched device path
    assert gp.log_likelihood.shape == (1,)
    np.testing.assert_allclose(gp.log_likelihood, g['ll3_chol'], rtol=1e-11)
    gp.compute_log_likelihood(g['hyp'], svd_method=True)           # GPU K + host SVD
    np.testing.assert_allclose(gp.log_likelihood, g['ll3_svd'], rtol=1e-9)
    gp.fit_nugget = True
    gp.compute_log_likelihood(g['hyp_fitnug'], svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, g['ll3_fitnug'], rtol=1e-11)
    g1 = cg.gaussian_process(ys[0], xs[0], kernel='RBF1D', y_err=es[0], substract_mean=True)
    g1.compute_log_likelihood(g['hyp'], svd_method=False)
    np.testing.assert_allclose(g1.log_likelihood, g['ll1_submean'], rtol=1e-11)
    ll = cg.log_likelihood_gp(ys[1], xs[1], 'RBF1D', g['hyp'], 0.04, y_err=es[1], svd_method=False)
    ref = orc.log_likelihood_gp(ys[1], xs[1], 'RBF1D', g['hyp'], 0.04, y_err=es[1], svd_method=False)
    np.testing.assert_allclose(ll, ref, rtol=1e-11)


def test_loglik_2d_golden(cg):
    g = load_golden('loglik_2d')
    gp = cg.gaussian_process_2d(g['y'], g['x'], y_err=g['e'])
    np.testing.assert_allclose(gp.hyperparameters, g['init_hyp'], rtol=1e-15)
    gp.compute_log_likelihood(g['hyp'], svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, g['ll_chol'], rtol=1e-11)


def test_loglik_not_pd_raises(cg):
    x = np.linspace(0, 5, 60)
    gp = cg.gaussian_process(np.sin(x), x)                         # noise-free: numerically singular
    with pytest.raises(np.linalg.LinAlgError):
        gp.compute_log_likelihood([1., 2.], svd_method=False)


# ----------------------------------------------------------------- a8 fit
def test_find_hyperparameters_golden(cg):
    g = load_golden('fit_1d_n60')
    gp = cg.gaussian_process(g['y'], g['x'], y_err=g['e'])
    np.testing.assert_allclose(gp.hyperparameters, g['init_guess'], rtol=1e-15)
    gp.find_hyperparameters(hyperparameter_guess=[0.8, 1.4], svd_method=False)
    np.testing.assert_allclose(gp.hyperparameters, g['fit_hyp'], rtol=1e-4)
    gn = cg.gaussian_process(g['y'], g['x'], y_err=g['e'])
    gn.find_hyperparameters(hyperparameter_guess=[0.8, 1.4], nugget=True, svd_method=False)
    np.testing.assert_allclose(gn.hyperparameters, g['fit_hyp_nug'], rtol=1e-3)
    assert abs(gn.nugget - float(g['fit_nugget'])) < 1e-3
    with pytest.raises(AssertionError):
        gp.find_hyperparameters(hyperparameter_guess=[1.], svd_method=False)


# ----------------------------------------------------------------- a10 / a11 prediction
def test_predict_c1_golden(cg):
    g = load_golden('c1_predict_1d_n256')
    ymax = np.abs(g['y']).max()
    gp = cg.gaussian_process(g['y'], g['x'], y_err=g['e'])
    gp.hyperparameters = g['hyp']
    gp.compute_log_likelihood(gp.hyperparameters, svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, g['loglik'], rtol=1e-11)
    gp.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gp.Prediction[0], g['pred'], rtol=0, atol=1e-9 * ymax)
    np.testing.assert_allclose(gp.variance[0], np.diag(g['cov']), rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.covariance_matrix[0], g['cov'], rtol=0, atol=1e-9)
    assert len(gp.covariance_matrix) == 1 and gp.as_the_same_time is False
    gp.get_prediction(new_binning=None, svd_method=False)
    np.testing.assert_allclose(gp.Prediction[0], g['pred_same'], rtol=0, atol=1e-9 * ymax)
    np.testing.assert_allclose(np.diag(gp.covariance_matrix[0]), g['cov_diag_same'], rtol=0, atol=1e-9)
    gp.get_prediction(new_binning=g['xnew'], svd_method=True)       # reference default route
    np.testing.assert_allclose(gp.Prediction[0], g['pred_svd'], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.diag(gp.covariance_matrix[0]), g['cov_diag_svd'], rtol=0, atol=1e-8)


def test_predict_with_mean_golden(cg):
    g = load_golden('predict_mean_1d_n80')
    hyp, ng = g['hyp'], float(g['nugget'])
    gm = cg.gaussian_process(g['y'], g['x'], y_err=g['e'], Mean_Y=g['mean_y'], Time_mean=g['mean_x'])
    gm.hyperparameters = hyp
    gm.nugget = ng
    gm.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gm.Prediction[0], g['pred'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gm.covariance_matrix[0], g['cov'], rtol=0, atol=1e-9)
    gm.get_prediction(new_binning=None, svd_method=False)
    np.testing.assert_allclose(gm.Prediction[0], g['pred_same'], rtol=0, atol=1e-9)
    gd = cg.gaussian_process(g['y'], g['x'], y_err=g['e'], Mean_Y=g['mean_y'], Time_mean=g['mean_x'], diff=[0.25])
    gd.hyperparameters = hyp
    gd.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gd.Prediction[0], g['pred_diff'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gd.variance[0], g['cov_diag_diff'], rtol=0, atol=1e-9)
    gs = cg.gaussian_process(g['y'], g['x'], y_err=g['e'], substract_mean=True)
    gs.hyperparameters = hyp
    gs.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gs.Prediction[0], g['pred_submean'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gs.variance[0], g['cov_diag_submean'], rtol=0, atol=1e-9)


def test_predict_2d_golden(cg):
    g = load_golden('predict_2d_n150')
    gp = cg.gaussian_process(g['y'], g['x'], kernel='RBF2D', y_err=g['e'])
    gp.hyperparameters = g['hyp']
    gp.nugget = float(g['nugget'])
    gp.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gp.Prediction[0], g['pred'], rtol=0, atol=1e-9 * np.abs(g['y']).max())
    np.testing.assert_allclose(gp.covariance_matrix[0], g['cov'], rtol=0, atol=1e-9)
    inv = gp.inv_kernel_matrix[0]
    np.testing.assert_allclose(inv, g['inv_kernel_matrix'], rtol=0, atol=1e-8 * np.abs(inv).max())
    K = gp.kernel_matrix[0]
    assert ulp_diff(K, orc.rbf_kernel_2d(g['x'], g['hyp'], nugget=float(g['nugget']), y_err=g['e'])) <= ULP


def test_predict_nobject_golden(cg):
    g = load_golden('predict_nobject_1d')
    xs, ys, es = objs(g)
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)       # ragged 31/44/25
    gp.hyperparameters = g['hyp']
    gp.get_prediction(new_binning=g['xnew'], svd_method=False)
    for i in range(3):
        np.testing.assert_allclose(gp.Prediction[i], g['pred_%d' % i], rtol=0, atol=1e-9)
        np.testing.assert_allclose(gp.variance[i], g['covdiag_%d' % i], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.diag(gp.covariance_matrix[i]), g['covdiag_%d' % i], rtol=0, atol=1e-9)
    # own grids: the reference's stale-index behaviour is reproducible on request ...
    import sys
    gpm = sys.modules['cosmogp_amd.gaussian_process']     # (the package re-exports a class of that name)
    xs, ys, es = objs(g, prefix='s')
    gq = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    gq.hyperparameters = g['hyp']
    try:
        gpm.REFERENCE_STALE_INDEX = True
        gq.get_prediction(new_binning=None, svd_method=False)
    finally:
        gpm.REFERENCE_STALE_INDEX = False
    for i in range(3):
        np.testing.assert_allclose(gq.Prediction[i], g['spred_stale_%d' % i], rtol=0, atol=1e-9)
    # ... and the default uses each object's own grid (oracle with the fix)
    gq.get_prediction(new_binning=None, svd_method=False)
    r = orc.predict(ys, xs, 'RBF1D', g['hyp'], 0., es, svd_method=False, fix_stale_index=True)
    for i in range(3):
        np.testing.assert_allclose(gq.Prediction[i], r['prediction'][i], rtol=0, atol=1e-9)
        np.testing.assert_allclose(gq.variance[i], np.diag(r['covariance'][i]), rtol=0, atol=1e-9)


def test_pull_golden(cg):
    g = load_golden('pull_1d_n20')
    bp = cg.build_pull([g['y']], [g['x']], g['hyp'], nugget=float(g['nugget']), y_err=[g['e']])
    bp.compute_pull(svd_method=False)
    np.testing.assert_allclose(bp.pull, g['pull'], rtol=0, atol=1e-8)
    np.testing.assert_allclose(bp.residual, g['residual'], rtol=0, atol=1e-8)
    assert abs(bp.pull_average - float(g['pull_average'])) < 1e-8
    assert abs(bp.pull_std - float(g['pull_std'])) < 1e-8
    bs = cg.build_pull([g['y']], [g['x']], g['hyp'], nugget=float(g['nugget']), y_err=[g['e']])
    bs.compute_pull(svd_method=False, substract_mean=True)
    np.testing.assert_allclose(bs.pull, g['pull_submean'], rtol=0, atol=1e-8)
    # chunked batches (tiny HBM budget: 3 sub-problems per launch sequence) give the same pulls
    import sys
    gpm = sys.modules['cosmogp_amd.gaussian_process']
    saved = gpm.BATCH_BYTES_LIMIT
    try:
        gpm.BATCH_BYTES_LIMIT = 3 * (128 + 128) * 128 * 8
        bc = cg.build_pull([g['y']], [g['x']], g['hyp'], nugget=float(g['nugget']), y_err=[g['e']])
        bc.compute_pull(svd_method=False)
    finally:
        gpm.BATCH_BYTES_LIMIT = saved
    assert np.array_equal(np.array(bc.pull), np.array(bp.pull))


# ----------------------------------------------------------------- fused / batched vs oracle
@pytest.mark.parametrize('n,m', [(1, 1), (2, 5), (127, 3), (128, 128), (129, 127), (513, 1), (1000, 300)])
def test_fit_shapes_vs_oracle(cg, n, m):
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(100 + n)
    x = np.sort(rng.uniform(0, 0.4 * n + 1, n))
    xn = rng.uniform(0, 0.4 * n + 1, m)
    e = rng.uniform(0.05, 0.1, n)
    y = np.sin(x) + 0.1 * rng.standard_normal(n)
    hyp = [0.8, 2.0]
    Kc = orc.rbf_kernel_1d(x, hyp, y_err=e, nugget=0.02)
    H = orc.rbf_kernel_1d(x, hyp, new_x=xn)
    mu, var, ll = orc.fit_predict_cho(Kc, y, H, np.full(m, hyp[0] ** 2 + 0.02 ** 2))
    dev = cg.default_device()
    dev.set_data(1, x, y)
    lld, ldd, mud, vard = dev.fit(K.kernel_params_1d(hyp), K.diag_term(n, e, 0., 0.02), xn, None, 0.02 ** 2)
    assert abs(lld - ll) <= 1e-11 * max(1., abs(ll))
    assert abs(ldd - np.linalg.slogdet(Kc)[1]) <= 1e-11 * max(1., abs(ldd))
    np.testing.assert_allclose(mud, mu, rtol=0, atol=1e-9 * max(1., np.abs(y).max()))
    np.testing.assert_allclose(vard, var, rtol=0, atol=1e-9)
    mu2, var2, cov2 = dev.predict(xn, 0.5, 0.02 ** 2, want_cov=True)      # re-use of the factor
    np.testing.assert_allclose(mu2, mu + 0.5, rtol=0, atol=1e-9 * max(1., np.abs(y).max()))
    np.testing.assert_allclose(np.diag(cov2), var, rtol=0, atol=1e-9)
    L = dev.fetch(1)
    np.testing.assert_allclose(L, np.linalg.cholesky(Kc), rtol=0, atol=1e-12)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))


def test_batch_ragged_vs_oracle(cg):
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(7)
    sizes = (200, 512, 300, 129, 512, 1, 77)
    xs = [np.sort(rng.uniform(-12, 42, k)) for k in sizes]
    es = [rng.uniform(0.02, 0.1, k) for k in sizes]
    ys = [np.sin(0.2 * a) + 0.05 * rng.standard_normal(len(a)) for a in xs]
    hyp, xn = [0.5, 8.0], np.linspace(-12, 42, 19)
    dev = cg.default_device()
    ll, info, mu, var = dev.batch_fit(1, xs, ys, [K.diag_term(len(a), e, 0., 0.) for a, e in zip(xs, es)],
                                      K.kernel_params_1d(hyp), xn, None, 0.)
    assert not info.any()
    for b in range(len(sizes)):
        Kc = orc.rbf_kernel_1d(xs[b], hyp, y_err=es[b])
        H = orc.rbf_kernel_1d(xs[b], hyp, new_x=xn)
        m0, v0, l0 = orc.fit_predict_cho(Kc, ys[b], H, np.full(19, hyp[0] ** 2))
        assert abs(l0 - ll[b]) <= 1e-11 * max(1., abs(l0))
        np.testing.assert_allclose(mu[b], m0, rtol=0, atol=1e-9)
        np.testing.assert_allclose(var[b], v0, rtol=0, atol=1e-9)
    # one bad object is reported per object and does not poison the others
    es_bad = [e.copy() for e in es]
    diags = [K.diag_term(len(a), e, 0., 0.) for a, e in zip(xs, es_bad)]
    diags[2] = np.full(sizes[2], -5.0)
    ll2, info2, _, _ = dev.batch_fit(1, xs, ys, diags, K.kernel_params_1d(hyp), xn, None, 0.)
    assert info2[2] > 0 and not info2[[0, 1, 3, 4, 5, 6]].any()
    np.testing.assert_allclose(ll2[[0, 1, 3, 4, 5, 6]], ll[[0, 1, 3, 4, 5, 6]], rtol=1e-13)


def test_sharded_equals_serial(cg):
    """Single-rank form of the multi-GPU path: blocks of objects fitted separately and
    concatenated equal one call over all objects (no cross-object coupling)."""
    from cosmogp_amd import kernel as K
    from cosmogp_amd import multigpu
    rng = np.random.default_rng(11)
    B = 12
    xs = [np.sort(rng.uniform(-12, 42, 256)) for _ in range(B)]
    ys = [np.sin(0.2 * a) + 0.05 * rng.standard_normal(256) for a in xs]
    ds = [np.full(256, 0.01) for _ in range(B)]
    kp, xn = K.kernel_params_1d([0.5, 8.0]), np.linspace(-12, 42, 19)
    dev = cg.default_device()
    ll, _, mu, var = dev.batch_fit(1, xs, ys, ds, kp, xn, None, 0.)
    for world in (2, 4, 8):
        parts = []
        for r in range(world):
            lo, hi = multigpu.shard_bounds(B, r, world)
            if hi > lo:
                parts.append(dev.batch_fit(1, xs[lo:hi], ys[lo:hi], ds[lo:hi], kp, xn, None, 0.))
        assert np.array_equal(np.concatenate([p[0] for p in parts]), ll)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), mu)
        assert np.array_equal(np.concatenate([p[3] for p in parts]), var)


# ----------------------------------------------------------------- full sizes of BASELINE.json
def _c2(rng):
    n, m = 4096, 1024
    x = np.sort(rng.uniform(0., 400., n))
    y = np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(n)
    return x, y, np.full(n, 0.1), rng.uniform(0., 400., m), [1., 2.]


def test_c2_full_size_vs_oracle(cg):
    """BASELINE config[1]: 1D N=4096, M=1024 -- direct comparison with the CPU oracle."""
    from cosmogp_amd import kernel as K
    x, y, e, xn, hyp = _c2(np.random.default_rng(2))
    Kc = orc.rbf_kernel_1d(x, hyp, y_err=e)
    H = orc.rbf_kernel_1d(x, hyp, new_x=xn)
    mu, var, ll = orc.fit_predict_cho(Kc, y, H, np.full(len(xn), 1.))
    dev = cg.default_device()
    dev.set_data(1, x, y)
    lld, _, mud, vard = dev.fit(K.kernel_params_1d(hyp), K.diag_term(len(x), e, 0., 0.), xn, None, 0.)
    assert abs(lld - ll) <= 1e-11 * abs(ll)
    np.testing.assert_allclose(mud, mu, rtol=0, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(vard, var, rtol=0, atol=1e-9)


def test_c3_full_size_properties(cg):
    """BASELINE config[2]: 2D N=8192, M=2500.  Size-independent properties:
    linearity of the mean in the residual, logdet(c K) = logdet K + N log c, 0 <= var <= prior,
    interpolation of a training point, and L L^T = K on random probes."""
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(3)
    n, m = 8192, 2500
    X = rng.uniform(0., 2048., (n, 2))
    g = np.linspace(20., 2028., 50)
    Xn = np.array([[a, b] for a in g for b in g])
    Xn[0] = X[17]                                             # one grid point on a training point
    y1 = np.sin(X[:, 0] / 150.) * np.cos(X[:, 1] / 170.) + 0.1 * rng.standard_normal(n)
    y2 = 0.3 * rng.standard_normal(n)
    hyp = [1., 120., 90., 500.]
    kp = K.kernel_params_2d(hyp)
    diag = K.diag_term(n, np.full(n, 0.1), 0., 0.)
    dev = cg.default_device()
    outs = []
    for r in (y1, y2, y1 + 2. * y2):
        dev.set_data(2, X, r)
        outs.append(dev.fit(kp, diag, Xn, None, 0.))
    (ll1, ld1, mu1, v1), (ll2, ld2, mu2, v2), (ll3, ld3, mu3, v3) = outs
    scale = np.abs(mu1).max() + 2 * np.abs(mu2).max()
    np.testing.assert_allclose(mu3, mu1 + 2. * mu2, rtol=0, atol=1e-9 * scale)     # linearity
    assert ld1 == ld2 == ld3 and np.array_equal(v1, v2)                            # K-only quantities
    assert v1.min() >= -1e-9 and v1.max() <= 1. + 1e-12                            # 0 <= var <= prior
    # a training point is reproduced to within its noise: |mu - y| small relative to prior
    assert abs(mu1[0] - y1[17]) < 0.5 and v1[0] < 0.02
    # log-likelihood identity: ll = -1/2 r'K^-1 r - n/2 log 2pi - 1/2 logdet, and quadratic form additivity
    q = lambda ll, ld: -2. * (ll + 0.5 * n * np.log(2 * np.pi) + 0.5 * ld)
    # (y1+2y2)'K^-1(y1+2y2) = q1 + 4 q2 + 4 y1'K^-1 y2 ; use symmetry check with (y1-2y2)
    dev.set_data(2, X, y1 - 2. * y2)
    ll4, ld4, _, _ = dev.fit(kp, diag, None)
    assert abs((q(ll3, ld3) + q(ll4, ld4)) - 2. * (q(ll1, ld1) + 4. * q(ll2, ld2))) <= 1e-9 * q(ll3, ld3)
    # scaling K by c (amp and diag) shifts logdet by n log c
    c = 3.0
    kpc = kp.copy()
    kpc[0] *= c
    lls, lds, _, _ = dev.fit(kpc, diag * c, None)            # c K: train amplitude kp[0] and the diagonal term
    assert abs(lds - (ld4 + n * np.log(c))) <= 1e-10 * abs(lds)
    # prediction AT training points: var = 1 - h'K^-1 h with h = K_row - diag e_i stays within the noise
    dev.set_data(2, X, y1)
    dev.fit(kp, diag, None)
    rows = rng.choice(n, 6, replace=False)
    mu_r, var_r, _ = dev.predict(X[rows], None, 0.)
    assert np.all(var_r >= -1e-9) and np.all(var_r < 0.02 + 1e-9)
    assert np.all(np.abs(mu_r - y1[rows]) < 0.5)


def test_c3_subsample_vs_oracle(cg):
    """Same 2D configuration at N=2048 (oracle finishes in seconds): direct comparison."""
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(33)
    n, m = 2048, 400
    X = rng.uniform(0., 1024., (n, 2))
    Xn = rng.uniform(0., 1024., (m, 2))
    y = np.sin(X[:, 0] / 150.) * np.cos(X[:, 1] / 170.) + 0.1 * rng.standard_normal(n)
    hyp = [1., 120., 90., 500.]
    e = np.full(n, 0.1)
    Kc = orc.rbf_kernel_2d(X, hyp, y_err=e)
    H = orc.rbf_kernel_2d(X, hyp, new_x=Xn)
    mu, var, ll = orc.fit_predict_cho(Kc, y, H, np.full(m, 1.))
    dev = cg.default_device()
    dev.set_data(2, X, y)
    lld, _, mud, vard = dev.fit(K.kernel_params_2d(hyp), K.diag_term(n, e, 0., 0.), Xn, None, 0.)
    # cond(K) ~ 1e4 here: tolerance scaled accordingly
    assert abs(lld - ll) <= 1e-10 * abs(ll)
    np.testing.assert_allclose(mud, mu, rtol=0, atol=1e-8 * np.abs(y).max())
    np.testing.assert_allclose(vard, var, rtol=0, atol=1e-8)


@pytest.mark.timeout(600)
def test_c3_and_c2_full_size_vs_oracle(cg):
    """BASELINE config[2] (2D N=8192, M=2500) and config[1] (1D N=4096, M=1024) at FULL size against the oracle directly
    (numpy / LAPACK on the host: ~10 s per object), through both product entry points: cgp_fit (one object, single launch)
    and cgp_batch_fit (two objects per call: the batched single-launch route, one leaf lane per object)."""
    from cosmogp_amd import kernel as K
    dev = cg.default_device()
    # ---- C3
    rng = np.random.default_rng(303)
    n, m = 8192, 2500
    g = np.linspace(20., 2028., 50)
    Xn = np.array([[a, b] for a in g for b in g])
    hyp = [1., 120., 90., 500.]
    objs = []
    for _ in range(2):
        X = rng.uniform(0., 2048., (n, 2))
        y = np.sin(X[:, 0] / 150.) * np.cos(X[:, 1] / 170.) + 0.1 * rng.standard_normal(n)
        objs.append((X, y))
    e = np.full(n, 0.1)
    kp, diag = K.kernel_params_2d(hyp), K.diag_term(n, e, 0., 0.)
    bll, binfo, bmu, bvar = dev.batch_fit(2, [o[0] for o in objs], [o[1] for o in objs], [diag, diag], kp, Xn, None, 0.)
    assert not binfo.any()
    for i, (X, y) in enumerate(objs):
        # tolerances COMPUTED from cond_1(K) of this matrix (conftest.oracle_fit_with_bounds), not asserted in a comment
        o = oracle_fit_with_bounds(orc.rbf_kernel_2d(X, hyp, y_err=e), y, orc.rbf_kernel_2d(X, hyp, new_x=Xn), np.full(m, 1.))
        assert_within_bounds(o, bll[i], bmu[i], bvar[i], 'C3 batch object %d' % i)
        assert abs(bll[i] - o['ll']) <= 1e-10 * abs(o['ll'])          # (the flat tolerances of rounds 1-3 stay as well)
        np.testing.assert_allclose(bmu[i], o['mu'], rtol=0, atol=1e-8 * np.abs(y).max())
        np.testing.assert_allclose(bvar[i], o['var'], rtol=0, atol=1e-8)
        if i == 0:
            dev.set_data(2, X, y)
            lls, _, mus, vars_ = dev.fit(kp, diag, Xn, None, 0.)
            assert_within_bounds(o, lls, mus, vars_, 'C3 single fit')
    # ---- C2
    n, m = 4096, 1024
    xn = np.linspace(0., 600., m)
    hyp1 = [1., 2.]
    objs = []
    for _ in range(2):
        x = np.sort(rng.uniform(0., 600., n))
        y = np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(n)
        objs.append((x, y))
    kp, diag = K.kernel_params_1d(hyp1), K.diag_term(n, e[:n], 0., 0.)
    bll, binfo, bmu, bvar = dev.batch_fit(1, [o[0] for o in objs], [o[1] for o in objs], [diag, diag], kp, xn, None, 0.)
    assert not binfo.any()
    for i, (x, y) in enumerate(objs):
        o = oracle_fit_with_bounds(orc.rbf_kernel_1d(x, hyp1, y_err=e[:n]), y, orc.rbf_kernel_1d(x, hyp1, new_x=xn), np.full(m, 1.))
        assert_within_bounds(o, bll[i], bmu[i], bvar[i], 'C2 batch object %d' % i)
        assert abs(bll[i] - o['ll']) <= 1e-10 * abs(o['ll'])
        np.testing.assert_allclose(bmu[i], o['mu'], rtol=0, atol=1e-8 * np.abs(y).max())
        np.testing.assert_allclose(bvar[i], o['var'], rtol=0, atol=1e-8)


def test_c4_batch_1024_lightcurves(cg):
    """BASELINE config[3]: 1024 x 1D N=512, M=19 in one batched launch sequence; a sample of
    objects is compared with the oracle, all of them are checked for sanity."""
    from cosmogp_amd import kernel as K
    B, n = 1024, 512
    xn = np.linspace(-12., 42., 19)
    xs, ys, ds, es = [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(4000 + i)
        x = np.sort(rng.uniform(-12., 42., n))
        e = rng.uniform(0.02, 0.1, n)
        xs.append(x)
        es.append(e)
        ys.append(0.5 * np.sin(x / 8.) + e * rng.standard_normal(n))
        ds.append(K.diag_term(n, e, 0., 0.))
    hyp = [0.5, 8.]
    dev = cg.default_device()
    ll, info, mu, var = dev.batch_fit(1, xs, ys, ds, K.kernel_params_1d(hyp), xn, None, 0.)
    assert not info.any() and np.all(np.isfinite(ll)) and np.all(var > -1e-9) and np.all(var <= 0.25 + 1e-12)
    for b in (0, 1, 511, 1023):
        Kc = orc.rbf_kernel_1d(xs[b], hyp, y_err=es[b])
        H = orc.rbf_kernel_1d(xs[b], hyp, new_x=xn)
        # (cond(K) ~ sigma^2 N / err^2 ~ 3e5: the bound is computed from the matrix, conftest.oracle_fit_with_bounds)
        o = oracle_fit_with_bounds(Kc, ys[b], H, np.full(19, 0.25))
        assert_within_bounds(o, ll[b], mu[b], var[b], 'C4 object %d' % b)
        assert abs(o['ll'] - ll[b]) <= 1e-9 * abs(o['ll'])            # (the flat tolerances of rounds 1-3 stay as well)
        np.testing.assert_allclose(mu[b], o['mu'], rtol=0, atol=1e-8)
        np.testing.assert_allclose(var[b], o['var'], rtol=0, atol=1e-8)


def test_eager_graph_lookahead_bit_identical(cg, monkeypatch):
    """The device-side fit is captured into a hipGraph (default); the graph with the opt-in two-stream
    look-ahead (CGP_LOOKAHEAD) and eager single-stream launches (CGP_NO_GRAPH) must agree.
    The look-ahead issues each update in column pieces, so a C element receives exactly the same
    products in the same order: results are bit-identical.  Each mode uses a fresh handle (graphs are
    cached)."""
    from cosmogp_amd import kernel as K
    from cosmogp_amd import _lib
    rng = np.random.default_rng(77)
    n, m = 1500, 333
    x = np.sort(rng.uniform(0, 600., n))
    xn = rng.uniform(0, 600., m)
    y = np.sin(0.05 * x) + 0.1 * rng.standard_normal(n)
    kp, dg = K.kernel_params_1d([1., 2.]), K.diag_term(n, np.full(n, 0.1), 0., 0.)
    outs = []
    monkeypatch.setenv('CGP_NO_DAG', '1')       # the launch-per-step route (the single-launch route: tests/test_dag.py)
    # (the opt-in left-looking panels give way to the look-ahead, and are within rounding of the default route, not bit-identical:
    # a suite run under CGP_LL=1, tools/r04_env_matrix.sh, compares the three modes on the default route like any other run)
    monkeypatch.delenv('CGP_LL', raising=False)
    for env in ({}, {'CGP_LOOKAHEAD': '1'}, {'CGP_NO_GRAPH': '1'}):
        for k in ('CGP_LOOKAHEAD', 'CGP_NO_GRAPH'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = _lib.Device(0)
        dev.set_data(1, x, y)
        first = dev.fit(kp, dg, xn, None, 0.)
        again = dev.fit(kp, dg, xn, None, 0.)           # replay of the cached graph
        assert first[0] == again[0] and np.array_equal(first[2], again[2]) and np.array_equal(first[3], again[3])
        outs.append(first)
        dev.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1]
        assert np.array_equal(o[2], outs[0][2]) and np.array_equal(o[3], outs[0][3])


def test_graph_replay_with_new_hyperparameters(cg):
    """A cached graph only points at HBM buffers: new kernel parameters / diagonal / residual through the
    same handle must give the new answer (find_hyperparameters relies on this)."""
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(78)
    n = 700
    x = np.sort(rng.uniform(0, 100., n))
    e = rng.uniform(0.05, 0.1, n)
    y = np.sin(0.3 * x) + e * rng.standard_normal(n)
    dev = cg.default_device()
    dev.set_data(1, x, y)
    for hyp, ng in (([1., 2.], 0.), ([0.7, 3.5], 0.05), ([1.3, 1.1], 0.0)):
        ll = dev.fit(K.kernel_params_1d(hyp), K.diag_term(n, e, 0., ng))[0]
        ref = orc.log_likelihood_gp(y, x, 'RBF1D', hyp, ng, y_err=e, svd_method=False)[0]
        assert abs(ll - ref) <= 1e-11 * abs(ref)
    dev.set_data(1, x, 2. * y)                               # new residual, same shape -> same graph
    ll2 = dev.fit(K.kernel_params_1d([1., 2.]), K.diag_term(n, e, 0., 0.))[0]
    ref2 = orc.log_likelihood_gp(2. * y, x, 'RBF1D', [1., 2.], 0., y_err=e, svd_method=False)[0]
    assert abs(ll2 - ref2) <= 1e-11 * abs(ref2)


def test_svd_method_optin_routes_to_cholesky_when_pd(cg):
    """Opt-in: svd_method=True (the reference default) tries the GPU Cholesky first; on PD input it agrees
    with the golden SVD answers, on singular input the device eigensolver takes over (never raises, like the reference)."""
    import sys
    gpm = sys.modules['cosmogp_amd.gaussian_process']
    g = load_golden('c1_predict_1d_n256')
    gp = cg.gaussian_process(g['y'], g['x'], y_err=g['e'])
    gp.hyperparameters = g['hyp']
    try:
        gpm.SVD_VIA_CHOLESKY_WHEN_PD = True
        gp.get_prediction(new_binning=g['xnew'], svd_method=True)
        np.testing.assert_allclose(gp.Prediction[0], g['pred_svd'], rtol=0, atol=1e-8)
        np.testing.assert_allclose(gp.variance[0], g['cov_diag_svd'], rtol=0, atol=1e-8)
        gp.compute_log_likelihood(g['hyp'], svd_method=True)
        np.testing.assert_allclose(gp.log_likelihood, g['loglik'], rtol=1e-10)
        x = np.linspace(0, 5, 60)
        gs = cg.gaussian_process(np.sin(x), x)                      # noise-free: singular K
        gs.hyperparameters = np.array([1., 2.])
        gs.get_prediction(new_binning=x[:5], svd_method=True)        # falls back, never raises
        assert np.all(np.isfinite(gs.Prediction[0]))
    finally:
        gpm.SVD_VIA_CHOLESKY_WHEN_PD = 'auto'


def test_nobject_fit_with_resident_batch(cg):
    """gaussian_process_nobject.find_hyperparameters: the objects are uploaded once (resident batch),
    every objective evaluation sends only the kernel-parameter block; result equals the oracle's
    Nelder-Mead run (same driver, same objective to 1e-11)."""
    g = load_golden('loglik_1d')
    xs, ys, es = objs(g)
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    dev = gp.device
    gen0 = dev.batch_generation
    gp.find_hyperparameters(hyperparameter_guess=[0.6, 7.], nugget=True, svd_method=False)
    assert dev.batch_generation == gen0 + 1                       # one upload for ~100+ evaluations
    y0 = orc.object_means(ys, xs)
    hyp, ng = orc.find_hyperparameters(ys, xs, 'RBF1D', [0.6, 7.], es, y0, fit_nugget=True, svd_method=False)
    np.testing.assert_allclose(gp.hyperparameters, hyp, rtol=1e-3)
    assert abs(gp.nugget - ng) < 1e-3
    # the likelihood at the optimum agrees tightly (same objective)
    gp.compute_log_likelihood(list(hyp) + [ng], svd_method=False)
    ref = orc.total_log_likelihood(ys, xs, 'RBF1D', hyp, ng, es, y0, svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, ref, rtol=1e-11)


def test_resident_diag_and_nugget_in_kp(cg):
    """cgp_fit with diag = NULL keeps the resident vector; kp[6] adds nugget^2 in the reference's order."""
    from cosmogp_amd import kernel as K
    rng = np.random.default_rng(91)
    n = 300
    x = np.sort(rng.uniform(0, 40., n))
    e = rng.uniform(0.05, 0.1, n)
    y = np.sin(x) + e * rng.standard_normal(n)
    dev = cg.default_device()
    dev.set_data(1, x, y)
    hyp = [0.9, 2.2]
    ll_a = dev.fit(K.kernel_params_1d(hyp, 0.07), K.diag_term(n, e, 0., 0.))[0]      # vector + kp[6]
    ll_b = dev.fit(K.kernel_params_1d(hyp, 0.07), None)[0]                             # resident vector
    ll_c = dev.fit(K.kernel_params_1d(hyp, 0.), K.diag_term(n, e, 0., 0.07))[0]       # all in the vector
    assert ll_a == ll_b == ll_c
    ref = orc.log_likelihood_gp(y, x, 'RBF1D', hyp, 0.07, y_err=e, svd_method=False)[0]
    assert abs(ll_a - ref) <= 1e-11 * abs(ref)
    dev.set_data(1, x, y)                                                              # resets the resident diagonal
    ll_z = dev.fit(K.kernel_params_1d(hyp, 0.3), None)[0]
    ref_z = orc.log_likelihood_gp(y, x, 'RBF1D', hyp, 0.3, y_err=None, svd_method=False)[0]
    assert abs(ll_z - ref_z) <= 1e-11 * abs(ref_z)


def test_sweep_lockstep(cg):
    """Hyper-parameter sweep: the candidates of a single-object GP are factored in lock-step by ONE launch
    sequence (cgp_sweep_loglik, shared data, one kernel-parameter block per candidate).  Every value is
    bit-identical to the candidate evaluated alone and equals the oracle to 1e-11; chunking by the HBM budget
    changes nothing; a non-positive-definite candidate raises like the serial loop."""
    from cosmogp_amd import multigpu
    rng = np.random.default_rng(55)
    n = 900
    X = rng.uniform(0., 300., (n, 2))
    y = np.sin(X[:, 0] / 40.) * np.cos(X[:, 1] / 50.) + 0.1 * rng.standard_normal(n)
    e = np.full(n, 0.1)
    gp = cg.gaussian_process_2d(y, X, y_err=e)
    table = np.array([[s_, l_, l_, 0.] for s_ in (0.7, 1.0, 1.4) for l_ in (20., 35., 60.)])
    a = multigpu.sweep_log_likelihood(gp, table)
    alone = []
    for row in table:
        gp.compute_log_likelihood(row, svd_method=False)
        alone.append(gp.log_likelihood[0])
    # (a candidate alone goes through the single-launch factorisation, a lock-step sequence through the launch-per-step
    # route: the same tiles with the sums associated differently, equal to rounding)
    np.testing.assert_allclose(a, np.array(alone), rtol=1e-12)
    for k in (0, 4, 8):
        ref = orc.log_likelihood_gp(y, X, 'RBF2D', table[k], 0., y_err=e, svd_method=False)[0]
        assert abs(a[k] - ref) <= 1e-11 * abs(ref)
    saved = multigpu.SWEEP_BYTES_LIMIT
    try:
        multigpu.SWEEP_BYTES_LIMIT = 4 * (1024 + 128) * 1024 * 8          # 4 candidates per launch sequence
        b = multigpu.sweep_log_likelihood(gp, table)
    finally:
        multigpu.SWEEP_BYTES_LIMIT = saved
    assert np.array_equal(a, b)
    # nugget column + fitted-nugget convention
    gp.fit_nugget = True
    tn = np.array([[1.0, 35., 35., 0., 0.05], [1.0, 35., 35., 0., 0.2]])
    c = multigpu.sweep_log_likelihood(gp, tn)
    ref = orc.log_likelihood_gp(y, X, 'RBF2D', tn[1, :4], 0.2, y_err=e, svd_method=False)[0]
    assert abs(c[1] - ref) <= 1e-11 * abs(ref)
    gp.fit_nugget = False
    # 1D, noise-free data: K is numerically singular for every candidate
    x1 = np.linspace(0., 5., 200)
    g1 = cg.gaussian_process(np.sin(x1), x1)
    with pytest.raises(np.linalg.LinAlgError):
        multigpu.sweep_log_likelihood(g1, np.array([[1., 2.], [1., 3.]]))


def test_concurrent_handles_independent_fits(cg):
    """Four host threads, four handles, different objects of different sizes in flight at once: every
    result is bit-identical to the same fit done alone (handles share nothing but the GPU)."""
    import threading
    from cosmogp_amd import kernel as K
    from cosmogp_amd import _lib
    rng = np.random.default_rng(321)
    jobs = []
    for n, m in ((1100, 90), (640, 33), (2048, 500), (129, 7), (1500, 260), (900, 1), (2300, 64), (512, 19)):
        x = np.sort(rng.uniform(0, 0.3 * n, n))
        jobs.append((x, np.sin(x) + 0.1 * rng.standard_normal(n), rng.uniform(0, 0.3 * n, m),
                     K.diag_term(n, np.full(n, 0.1), 0., 0.)))
    kp = K.kernel_params_1d([0.9, 2.0], 0.02)
    solo = cg.default_device()
    ref = []
    for x, y, xn, dg in jobs:
        solo.set_data(1, x, y)
        ref.append(solo.fit(kp, dg, xn, None, 0.02 ** 2))
    devs = _lib.device_pool(4)
    out = [None] * len(jobs)

    def work(slot):
        for rep in range(3):                       # repeat: graph replay paths too
            for k in range(slot, len(jobs), 4):
                x, y, xn, dg = jobs[k]
                devs[slot].set_data(1, x, y)
                out[k] = devs[slot].fit(kp, dg, xn, None, 0.02 ** 2)

    ths = [threading.Thread(target=work, args=(s_,)) for s_ in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for o, r in zip(out, ref):
        assert o[0] == r[0] and o[1] == r[1] and np.array_equal(o[2], r[2]) and np.array_equal(o[3], r[3])


def test_n32768_two_cluster_consistency(cg):
    """Beyond BASELINE's largest size (index arithmetic past 2^31 bytes per matrix: 8.6 GB at N = 32768): two point
    clusters so far apart that every cross-covariance underflows to exactly 0 make K block diagonal, so the
    log-likelihood of the whole is the sum of the parts (each computed by an N = 16384 fit) and the predictions
    near one cluster equal that cluster's own."""
    rng = np.random.default_rng(77)
    n = 16384
    xa = np.sort(rng.uniform(0., 3000., n))
    xb = np.sort(rng.uniform(1e6, 1e6 + 3000., n))
    f = lambda x: np.sin(x / 7.) + 0.3 * np.cos(x / 1.3)
    ya = f(xa) + 0.1 * rng.standard_normal(n)
    yb = f(xb - 1e6) + 0.1 * rng.standard_normal(n)
    e = np.full(n, 0.1)
    hyp = [1.0, 2.0]
    grid = np.linspace(100., 2900., 257)
    ga = cg.gaussian_process(ya, xa, kernel='RBF1D', y_err=e)
    ga.hyperparameters = hyp
    ga.compute_log_likelihood(hyp, svd_method=False)
    lla = ga.log_likelihood[0]
    ga.get_prediction(new_binning=grid, COV=False, svd_method=False)
    gb = cg.gaussian_process(yb, xb, kernel='RBF1D', y_err=e)
    gb.compute_log_likelihood(hyp, svd_method=False)
    llb = gb.log_likelihood[0]
    gab = cg.gaussian_process(np.concatenate([ya, yb]), np.concatenate([xa, xb]), kernel='RBF1D',
                              y_err=np.concatenate([e, e]))
    gab.hyperparameters = hyp
    gab.compute_log_likelihood(hyp, svd_method=False)
    assert abs(gab.log_likelihood[0] - (lla + llb)) <= 1e-11 * abs(lla + llb)
    gab.get_prediction(new_binning=grid, COV=False, svd_method=False)
    np.testing.assert_allclose(gab.Prediction[0], ga.Prediction[0], rtol=0, atol=1e-10)
    np.testing.assert_allclose(gab.variance[0], ga.variance[0], rtol=0, atol=1e-10)


def test_nobject_mixed_sizes_are_bucketed(cg):
    """One large object among small ones: objects are bucketed by padded size (three launch sequences here, not one
    padded to 1024), results in object order, equal to every object fitted alone bit for bit."""
    import sys
    rng = np.random.default_rng(17)
    sizes = [30, 900, 45, 200, 130, 60]
    xs = [np.sort(rng.uniform(0., 0.3 * n + 5., n)) for n in sizes]
    es = [rng.uniform(0.05, 0.2, n) for n in sizes]
    ys = [np.sin(x / 3.) + e * rng.standard_normal(len(x)) for x, e in zip(xs, es)]
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    hyp = [0.9, 2.5]
    gp.hyperparameters = hyp
    chunks = gp._batch_chunks(list(range(6)))
    assert sorted(map(sorted, chunks)) == [[0, 2, 5], [1], [3, 4]]
    grid = np.linspace(0., 20., 33)
    gp.get_prediction(new_binning=grid, svd_method=False)
    gp.compute_log_likelihood(hyp, svd_method=False)
    total = 0
    for i in range(6):
        g1 = cg.gaussian_process(ys[i], xs[i], kernel='RBF1D', y_err=es[i])
        g1.hyperparameters = hyp
        g1.get_prediction(new_binning=grid, svd_method=False)
        # (bit for bit for objects of one 128-block; larger single objects take the single-launch route: equal to rounding)
        np.testing.assert_allclose(g1.Prediction[0], gp.Prediction[i], rtol=0, atol=1e-11)
        np.testing.assert_allclose(g1.variance[0], gp.variance[i], rtol=0, atol=1e-11)
        g1.compute_log_likelihood(hyp, svd_method=False)
        total = total + g1.log_likelihood
    np.testing.assert_allclose(total, gp.log_likelihood, rtol=1e-12)
    gp.get_prediction(new_binning=None, svd_method=False)            # own grids, bucketed too
    want = orc.predict(ys, xs, 'RBF1D', hyp, 0., es, new_binning=None, svd_method=False)
    for i in range(6):
        np.testing.assert_allclose(gp.Prediction[i], want['prediction'][i], rtol=0, atol=1e-9)


def test_shared_then_own_grids_same_shape(cg):
    """Same batch, same number of prediction points, first one grid shared by all objects and then one grid per
    object: the two launch graphs must not be confused (they differ only in a stride)."""
    rng = np.random.default_rng(23)
    n = 64
    xs = [np.sort(rng.uniform(0., 20., n)) for _ in range(3)]
    es = [np.full(n, 0.1) for _ in range(3)]
    ys = [np.sin(x / 2.) + 0.1 * rng.standard_normal(n) for x in xs]
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    hyp = [1.0, 2.0]
    gp.hyperparameters = hyp
    shared = np.linspace(0., 20., n)                     # m == n: the own-grid call below has the same shapes
    for _ in range(2):
        gp.get_prediction(new_binning=shared, COV=False, svd_method=False)
        want = orc.predict(ys, xs, 'RBF1D', hyp, 0., es, new_binning=shared, svd_method=False, cov=False)
        for i in range(3):
            np.testing.assert_allclose(gp.Prediction[i], want['prediction'][i], rtol=0, atol=1e-10)
        gp.get_prediction(new_binning=None, COV=False, svd_method=False)
        want = orc.predict(ys, xs, 'RBF1D', hyp, 0., es, new_binning=None, svd_method=False, cov=False)
        for i in range(3):
            np.testing.assert_allclose(gp.Prediction[i], want['prediction'][i], rtol=0, atol=1e-10)


def test_pull_many_objects_default_arguments(cg):
    """build_pull over several objects of different sizes with the reference's default arguments
    (svd_method=True): all leave-one-out sub-problems go down as bucketed batches (no per-point loop) and equal the
    oracle's reference loop; noise-free data take the eigensolver route and still give finite pulls."""
    rng = np.random.default_rng(31)
    sizes = [12, 35, 150, 20, 131]
    xs = [np.sort(rng.uniform(0., 0.5 * n + 5., n)) for n in sizes]
    es = [rng.uniform(0.05, 0.2, n) for n in sizes]
    ys = [np.sin(x / 3.) + e * rng.standard_normal(len(x)) for x, e in zip(xs, es)]
    hyp = [0.9, 2.5]
    dev = cg._lib.Device()
    calls = []
    real = dev.batch_fit
    dev.batch_fit = lambda *a, **k: (calls.append(len(a[1])), real(*a, **k))[1]
    import sys
    pm = sys.modules['cosmogp_amd.pull']
    want_pull, want_res, _ = orc.compute_pull(ys, xs, hyp, nugget=0.03, y_err=es, svd_method=True)
    # default: every object's leave-one-out predictions from ONE factorisation of its full matrix (cgp_batch_loo)
    bp = cg.build_pull(ys, xs, hyp, nugget=0.03, y_err=es, device=dev)
    bp.compute_pull()                                            # svd_method=True, substract_mean=False
    assert calls == []
    np.testing.assert_allclose(bp.pull, want_pull, rtol=0, atol=1e-7)
    np.testing.assert_allclose(bp.residual, want_res, rtol=0, atol=1e-8)
    # the reference's own arithmetic: every (object, left-out point) pair refitted, as bucketed batches
    try:
        pm.LOO_BY_REFITTING = True
        bl = cg.build_pull(ys, xs, hyp, nugget=0.03, y_err=es, device=dev)
        bl.compute_pull()
    finally:
        pm.LOO_BY_REFITTING = False
    assert sorted(calls) == sorted([12 + 35 + 20, 150 + 131])    # two buckets (Np = 128, 256), every sub-problem once
    np.testing.assert_allclose(bl.pull, want_pull, rtol=0, atol=1e-7)
    np.testing.assert_allclose(bl.residual, want_res, rtol=0, atol=1e-8)
    np.testing.assert_allclose(bp.pull, bl.pull, rtol=0, atol=1e-9)
    np.testing.assert_allclose(bp.residual, bl.residual, rtol=0, atol=1e-10)
    bm = cg.build_pull(ys[:2], xs[:2], hyp, nugget=0.03, y_err=es[:2], device=dev)
    bm.compute_pull(substract_mean=True)
    want_pull, _, _ = orc.compute_pull(ys[:2], xs[:2], hyp, nugget=0.03, y_err=es[:2], svd_method=True, substract_mean=True)
    np.testing.assert_allclose(bm.pull, want_pull, rtol=0, atol=1e-7)
    x0 = np.linspace(0., 4., 12)
    b0 = cg.build_pull([np.sin(x0)], [x0], [1., 1.], device=dev)          # no y_err, no nugget: singular sub-matrices
    b0.compute_pull()
    assert len(b0.pull) == 12 and np.all(np.isfinite(b0.residual))


def test_pull_from_full_factor_equals_refitting(cg):
    """Leave-one-out predictions from one factorisation per object (cgp_batch_loo, the default of build_pull) against the
    literal refits, for everything that enters the identities: 1D and 2D kernels (the 2D train matrix carries no amplitude,
    cross rows do; lxy != 0), a nugget, a mean function with the per-subset offset of mean.py:95-97, a given `diff`,
    substract_mean, objects of several padded sizes in one call."""
    import sys
    pm = sys.modules['cosmogp_amd.pull']
    rng = np.random.default_rng(77)

    def both(make):
        outs = []
        for literal in (False, True):
            try:
                pm.LOO_BY_REFITTING = literal
                bp = make()
            finally:
                pm.LOO_BY_REFITTING = False
            outs.append((np.array(bp.pull), np.array(bp.residual)))
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=0, atol=2e-9)
        np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=2e-10)
        return outs[0]

    sizes = [9, 140, 60]
    xs = [np.sort(rng.uniform(0., 0.6 * n + 4., n)) for n in sizes]
    es = [rng.uniform(0.05, 0.2, n) for n in sizes]
    ys = [1.5 + np.sin(x / 3.) + e * rng.standard_normal(len(x)) for x, e in zip(xs, es)]
    gx = np.linspace(-5., 100., 60)
    gm = 1.5 + 0.8 * np.sin(gx / 3.)

    def run(**kw):
        args = dict(kw)
        cp = {k: args.pop(k) for k in ('diff', 'substract_mean', 'svd_method') if k in args}
        def make():
            bp = cg.build_pull(ys, xs, [0.9, 2.5], y_err=es, **args)
            bp.compute_pull(**cp)
            return bp
        return both(make)

    run(nugget=0.05)
    run(nugget=0.05, substract_mean=True)
    run(nugget=0.0, y_mean=gm, x_axis_mean=gx)
    run(nugget=0.02, y_mean=gm, x_axis_mean=gx, diff=[0.3, -0.1, 0.2], svd_method=False)
    # 2D
    sizes2 = [30, 130]
    X = [rng.uniform(0., 40., (n, 2)) for n in sizes2]
    E = [rng.uniform(0.05, 0.15, n) for n in sizes2]
    Y = [np.sin(x[:, 0] / 6.) * np.cos(x[:, 1] / 7.) + e * rng.standard_normal(len(x)) for x, e in zip(X, E)]

    def make2():
        bp = cg.build_pull(Y, X, [0.8, 6.0, 7.5, 9.0], nugget=0.03, y_err=E, kernel='RBF2D')
        bp.compute_pull()
        return bp
    pull2, res2 = both(make2)
    want_pull, want_res, _ = orc.compute_pull(Y, X, [0.8, 6.0, 7.5, 9.0], nugget=0.03, y_err=E, kernel='RBF2D', svd_method=True)
    np.testing.assert_allclose(pull2, want_pull, rtol=0, atol=1e-7)
    np.testing.assert_allclose(res2, want_res, rtol=0, atol=1e-8)
    # one object of 3000 points: 3000 refits of 2999 points are out of reach, one factorisation takes milliseconds
    n = 3000
    x = np.sort(rng.uniform(0., 0.4 * n, n))
    e = np.full(n, 0.1)
    y = np.sin(x / 3.) + e * rng.standard_normal(n)
    bp = cg.build_pull([y], [x], [1.0, 2.0], nugget=0.0, y_err=[e])
    bp.compute_pull()
    assert len(bp.pull) == n and abs(bp.pull_std - 1.) < 0.1 and abs(bp.pull_average) < 0.1
    t = 1234                                            # one of them against the oracle's refit
    keep = np.ones(n, dtype=bool); keep[t] = False
    Kc = orc.rbf_kernel_1d(x[keep], [1., 2.], y_err=e[keep])
    mu, var, _ = orc.fit_predict_cho(Kc, y[keep], orc.rbf_kernel_1d(x[keep], [1., 2.], new_x=x[t:t + 1]), np.full(1, 1.))
    assert abs(bp.residual[t] - (mu[0] - y[t])) < 1e-9
    assert abs(bp.pull[t] - (mu[0] - y[t]) / np.sqrt(e[t] ** 2 + abs(var[0]))) < 1e-8


def test_abi_status_codes(cg):
    """Raw C-ABI calls: call-order and argument errors come back as negative status codes with a message (no
    crash, no silent success); a non-positive-definite matrix as the positive LAPACK-style info."""
    import ctypes
    from cosmogp_amd import _lib
    lib = _lib.load_library()
    dp = ctypes.POINTER(ctypes.c_double)
    h = ctypes.c_void_p()
    assert lib.cgp_create(0, ctypes.byref(h)) == 0
    try:
        kp = (ctypes.c_double * 7)(1., 1., 4., 0., 0., 0., 0.)
        ll = ctypes.c_double()
        # nothing uploaded yet
        assert lib.cgp_fit(h, kp, None, None, 0, None, 0., ctypes.byref(ll), None, None, None) == -3
        assert b'no data' in lib.cgp_last_error(h)
        five = (ctypes.c_double * 5)()
        assert lib.cgp_fit_grad(h, kp, None, ctypes.byref(ll), None, five) == -3
        assert lib.cgp_sweep_loglik(h, 1, kp, None, ctypes.byref(ll), None) == -3
        assert lib.cgp_batch_fit_grad(h, kp, ctypes.byref(ll), None, five) == -3
        assert lib.cgp_fit_predict(h, kp, 1, None, 0., five, None, None) == -3          # no factor
        assert lib.cgp_bench_fit_objects(h, 1, 1, ctypes.byref(ll)) == -3
        # bad arguments
        x = (ctypes.c_double * 4)(0., 1., 2., 3.)
        assert lib.cgp_fit_set_data(h, 3, x, x, 4) == -1
        assert lib.cgp_fit_set_data(h, 1, x, x, 0) == -1
        assert lib.cgp_rbf(h, 1, x, 4, None, 0, kp, None, None, 4) == -1
        assert lib.cgp_fit_set_data(h, 1, x, x, 4) == 0
        assert lib.cgp_sweep_loglik(h, 0, kp, None, ctypes.byref(ll), None) == -1
        assert lib.cgp_sweep_loglik(h, 257, kp, None, ctypes.byref(ll), None) == -1
        assert lib.cgp_fit(h, None, None, None, 0, None, 0., ctypes.byref(ll), None, None, None) == -1
        # more objects than a grid dimension holds (both routes of cgp_batch_fit: the pipelined one checks for itself)
        nbig = 65536
        ns = (ctypes.c_int64 * nbig)(*([4] * nbig))
        xb = (ctypes.c_double * (4 * nbig))()
        llb = (ctypes.c_double * nbig)()
        assert lib.cgp_batch_set_data(h, 1, nbig, ns, xb, xb, None) == -1 and b'65535' in lib.cgp_last_error(h)
        assert lib.cgp_batch_fit(h, 1, nbig, ns, xb, xb, None, kp, None, 0, 0, None, 0., llb, None, None, None) == -1
        assert b'65535' in lib.cgp_last_error(h)
        ns65 = (ctypes.c_int64 * nbig)(*([65] * nbig))            # (too long to be packed: the pipelined route)
        x65 = (ctypes.c_double * (65 * nbig))()
        assert lib.cgp_batch_fit(h, 1, nbig, ns65, x65, x65, None, kp, None, 0, 0, None, 0., llb, None, None, None) == -1
        assert b'65535' in lib.cgp_last_error(h)
        del x65, xb
        # a valid call still works on the same handle afterwards ...
        diag = (ctypes.c_double * 4)(0.1, 0.1, 0.1, 0.1)
        assert lib.cgp_fit(h, kp, diag, None, 0, None, 0., ctypes.byref(ll), None, None, None) == 0
        ref = orc.log_likelihood_gp(np.array([0., 1., 2., 3.]), np.array([0., 1., 2., 3.]), 'RBF1D', [1., 2.], 0.,
                                    y_err=np.sqrt(np.full(4, 0.1)), svd_method=False)[0]
        assert abs(ll.value - ref) <= 1e-12 * abs(ref)
        # ... and a matrix that is not positive definite reports the failing leading minor
        K = np.eye(5)
        K[3, 3] = -2.
        inv = np.empty((5, 5))
        assert lib.cgp_cholesky_inverse(h, K.ctypes.data_as(dp), 5, 5, inv.ctypes.data_as(dp), None) == 4
    finally:
        assert lib.cgp_destroy(h) == 0
    assert lib.cgp_destroy(None) == -1 and lib.cgp_synchronize(None) == -1
