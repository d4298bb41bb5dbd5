"""The reference author's own cross-check (tests/old/compare_cosmogp_george.py:77-99, tests/old/test_gp_algorithm.py:328-350):
a textbook cho_solve GP equals scikit-learn's GaussianProcessRegressor with the same kernel and noise.  SURVEY.md 8(c) names
scikit-learn as the independent second oracle available in this image; the reference's import path
(`sklearn.gaussian_process.gpr`) is obsolete, `sklearn.gaussian_process` is today's.  CPU only: this pins the ORACLE
(oracle.fit_predict_cho, oracle.predict, oracle.total_log_likelihood -- the checkers of every GPU parity test) against an
implementation that shares no code with either the reference or this repository."""
import numpy as np
import pytest

from oracle import cosmogp_oracle as orc

sk = pytest.importorskip('sklearn.gaussian_process')
from sklearn.gaussian_process.kernels import RBF, ConstantKernel  # noqa: E402


def _data(seed, n=120, m=40):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0., 10., n))
    e = rng.uniform(0.05, 0.2, n)
    y = 0.8 * np.sin(x) + e * rng.standard_normal(n)
    return x, y, e, np.linspace(-0.5, 10.5, m)


@pytest.mark.parametrize('seed,sigma,ell', [(0, 0.8, 1.3), (1, 1.7, 0.6), (2, 0.3, 3.0)])
def test_oracle_equals_sklearn_gpr_1d(seed, sigma, ell):
    x, y, e, xn = _data(seed)
    gpr = sk.GaussianProcessRegressor(kernel=ConstantKernel(sigma ** 2, 'fixed') * RBF(ell, 'fixed'), alpha=e ** 2,
                                      optimizer=None, normalize_y=False)
    gpr.fit(x[:, None], y)
    mu_s, cov_s = gpr.predict(xn[:, None], return_cov=True)
    ll_s = gpr.log_marginal_likelihood_value_

    # the textbook form on the oracle's own kernel matrices (kernel.py:71-75)
    K = orc.rbf_kernel_1d(x, [sigma, ell], y_err=e)
    H = orc.rbf_kernel_1d(x, [sigma, ell], new_x=xn)
    mu, var, ll = orc.fit_predict_cho(K, y, H, np.full(len(xn), sigma ** 2))
    np.testing.assert_allclose(mu, mu_s, rtol=0, atol=1e-10 * np.abs(y).max())
    np.testing.assert_allclose(var, np.diag(cov_s), rtol=0, atol=1e-10 * sigma ** 2)
    assert abs(ll - ll_s) <= 1e-11 * abs(ll_s)

    # and the reference's own route (explicit inverse, Gaussian_process.py:270-361, :13-75) as the oracle restates it
    ref = orc.predict([y], [x], 'RBF1D', [sigma, ell], 0., [e], new_binning=xn, svd_method=False)
    np.testing.assert_allclose(ref['prediction'][0], mu_s, rtol=0, atol=1e-8 * np.abs(y).max())
    np.testing.assert_allclose(ref['covariance'][0], cov_s, rtol=0, atol=1e-8 * sigma ** 2)
    lls = orc.total_log_likelihood([y], [x], 'RBF1D', [sigma, ell], 0., [e], [0.], svd_method=False)
    assert abs(lls[0] - ll_s) <= 1e-10 * abs(ll_s)
    # the reference's default (svd_method=True) gives the same on this positive definite matrix
    llv = orc.total_log_likelihood([y], [x], 'RBF1D', [sigma, ell], 0., [e], [0.], svd_method=True)
    assert abs(llv[0] - ll_s) <= 1e-9 * abs(ll_s)


def test_oracle_equals_sklearn_gpr_2d_isotropic():
    """2D: the reference's train-train matrix omits sigma^2 (kernel.py:146), so the identity holds for sigma = 1 (where the quirk is
    invisible) with an isotropic length scale (hyperparameters [1, l, l, 0] enter as l^2 on the diagonal of L, kernel.py:127-130)."""
    rng = np.random.default_rng(7)
    n, m, ell = 150, 30, 2.5
    X = rng.uniform(0., 10., (n, 2))
    e = rng.uniform(0.05, 0.15, n)
    y = np.sin(X[:, 0]) * np.cos(X[:, 1] / 2.) + e * rng.standard_normal(n)
    Xn = rng.uniform(0., 10., (m, 2))
    gpr = sk.GaussianProcessRegressor(kernel=RBF(ell, 'fixed'), alpha=e ** 2, optimizer=None).fit(X, y)
    mu_s, cov_s = gpr.predict(Xn, return_cov=True)
    ref = orc.predict([y], [X], 'RBF2D', [1., ell, ell, 0.], 0., [e], new_binning=Xn, svd_method=False)
    np.testing.assert_allclose(ref['prediction'][0], mu_s, rtol=0, atol=1e-8 * np.abs(y).max())
    np.testing.assert_allclose(ref['covariance'][0], cov_s, rtol=0, atol=1e-8)
    lls = orc.total_log_likelihood([y], [X], 'RBF2D', [1., ell, ell, 0.], 0., [e], [0.], svd_method=False)
    assert abs(lls[0] - gpr.log_marginal_likelihood_value_) <= 1e-10 * abs(lls[0])
