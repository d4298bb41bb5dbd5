"""Analytic gradient of the log marginal likelihood (an extension -- the
reference only searches derivative-free, Gaussian_process.py:247, so it holds no golden for it).
Pinning chain: reference goldens -> oracle.log_likelihood_gp (tests/test_oracle_golden.py) -> central
differences of it -> oracle.log_likelihood_gradient -> cgp_fit_grad on the GPU."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import cosmogp_oracle as orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _data_1d(n=70, seed=11):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0., 20., n))
    e = rng.uniform(0.05, 0.2, n)
    y = np.sin(x / 2.5) + e * rng.standard_normal(n)
    return x, y, e


def _data_2d(n=90, seed=12):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0., 30., (n, 2))
    e = rng.uniform(0.05, 0.15, n)
    y = np.sin(x[:, 0] / 6.) * np.cos(x[:, 1] / 7.) + e * rng.standard_normal(n)
    return x, y, e


def _fd(fun, theta, eps=1e-6):
    theta = np.asarray(theta, dtype=np.float64)
    out = np.zeros(len(theta))
    for k in range(len(theta)):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        out[k] = (fun(tp) - fun(tm)) / (2. * eps)
    return out


def test_oracle_gradient_matches_central_differences_1d():
    x, y, e = _data_1d()
    theta = np.array([0.9, 2.2, 0.07])

    def ll(t):
        return orc.log_likelihood_gp(y, x, 'RBF1D', t[:2], t[2], y_err=e, y_mean=0.1, svd_method=False)[0]

    g = orc.log_likelihood_gradient(y, x, 'RBF1D', theta[:2], theta[2], y_err=e, y_mean=0.1)
    np.testing.assert_allclose(g, _fd(ll, theta), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize('s2', [False, True])
def test_oracle_gradient_matches_central_differences_2d(s2):
    x, y, e = _data_2d()
    theta = np.array([0.8, 6.0, 7.5, 9.0, 0.05])

    def ll(t):
        k = lambda xx, h, nugget=0., y_err=None: orc.rbf_kernel_2d(xx, h, nugget=nugget, y_err=y_err, sigma2_in_train=s2)
        return orc.log_likelihood_gp(y, x, k, t[:4], t[4], y_err=e, y_mean=0., svd_method=False)[0]

    g = orc.log_likelihood_gradient(y, x, 'RBF2D', theta[:4], theta[4], y_err=e, y_mean=0., sigma2_in_train=s2)
    np.testing.assert_allclose(g, _fd(ll, theta), rtol=2e-6, atol=1e-6)
    if not s2:
        assert g[0] == 0.          # the reference's train-train matrix has no sigma^2 (kernel.py:146)


def test_host_chain_rule_with_fake_device():
    """cosmogp_amd's host side (moment sums -> gradient, summing objects, nugget column, L-BFGS-B driver)
    on the oracle-backed device stand-in."""
    import cosmogp_amd
    from fake_device import FakeDevice
    x, y, e = _data_1d()
    gp = cosmogp_amd.gaussian_process(y, x, kernel='RBF1D', y_err=e, device=FakeDevice())
    gp.nugget = 0.07
    g = gp.compute_log_likelihood_gradient([0.9, 2.2])
    ref = orc.log_likelihood_gradient(y, x, 'RBF1D', [0.9, 2.2], 0.07, y_err=e, y_mean=0.)
    np.testing.assert_allclose(g, ref[:2], rtol=1e-9)
    np.testing.assert_allclose(gp.log_likelihood, orc.log_likelihood_gp(y, x, 'RBF1D', [0.9, 2.2], 0.07, y_err=e,
                                                                        y_mean=0., svd_method=False), rtol=1e-12)
    gp.fit_nugget = True
    g = gp.compute_log_likelihood_gradient([0.9, 2.2, 0.07])
    np.testing.assert_allclose(g, ref, rtol=1e-9)
    # 2D, three objects: gradients add up
    x2, y2, e2 = _data_2d()
    cuts = [(0, 30), (30, 55), (55, 90)]
    xs, ys, es = ([a[lo:hi] for lo, hi in cuts] for a in (x2, y2, e2))
    gq = cosmogp_amd.gaussian_process_nobject(ys, xs, kernel='RBF2D', y_err=es, device=FakeDevice())
    h = [0.8, 6.0, 7.5, 9.0]
    g = gq.compute_log_likelihood_gradient(h)
    ref = sum(orc.log_likelihood_gradient(ys[i], xs[i], 'RBF2D', h, 0., y_err=es[i], y_mean=0.) for i in range(3))
    np.testing.assert_allclose(g, ref[:4], rtol=1e-9, atol=1e-12)
    assert gq.device.calls.count('batch_fit_grad:3') == 1          # the three objects went down in one launch sequence
    # the gradient-based search lands on the optimum of the derivative-free one (golden from the reference)
    gg = load_golden('fit_1d_n60')
    gp = cosmogp_amd.gaussian_process(gg['y'], gg['x'], kernel='RBF1D', y_err=gg['e'], device=FakeDevice())
    gp.find_hyperparameters(hyperparameter_guess=list(gg['guess']), svd_method=False, method='L-BFGS-B')
    np.testing.assert_allclose(gp.hyperparameters, gg['fit_hyp'], rtol=2e-3)
    assert gp.fit_result.nfev < 40


# ----------------------------------------------------------------------------------- GPU
@pytest.fixture(scope='module')
def cg():
    import cosmogp_amd
    cosmogp_amd._lib.default_device()
    return cosmogp_amd


@pytest.mark.gpu
def test_gpu_gradient_1d_vs_oracle(cg):
    x, y, e = _data_1d(n=333)
    gp = cg.gaussian_process(y, x, kernel='RBF1D', y_err=e)
    gp.fit_nugget = True
    theta = [0.9, 2.2, 0.07]
    g = gp.compute_log_likelihood_gradient(theta)
    ref = orc.log_likelihood_gradient(y, x, 'RBF1D', theta[:2], theta[2], y_err=e, y_mean=0.)
    np.testing.assert_allclose(g, ref, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(gp.log_likelihood, orc.log_likelihood_gp(y, x, 'RBF1D', theta[:2], theta[2], y_err=e,
                                                                        y_mean=0., svd_method=False), rtol=1e-11)
    # the likelihood of the gradient call is the one of the plain call (to rounding: the plain call of a single object
    # larger than one block takes the single-launch factorisation, the gradient sweep the launch-per-step route)
    ll = gp.log_likelihood.copy()
    gp.compute_log_likelihood(theta, svd_method=False)
    np.testing.assert_allclose(ll, gp.log_likelihood, rtol=1e-12)
    # and the factor it leaves behind serves predictions
    gp.hyperparameters, gp.nugget, gp.fit_nugget = np.array(theta[:2]), theta[2], False
    gp.get_prediction(new_binning=np.linspace(0., 20., 50), svd_method=False)
    want = orc.predict([y], [x], 'RBF1D', theta[:2], theta[2], [e], new_binning=np.linspace(0., 20., 50),
                       svd_method=False)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('s2', [False, True])
def test_gpu_gradient_2d_vs_oracle(cg, s2):
    mod = sys.modules['cosmogp_amd.kernel']
    x, y, e = _data_2d(n=300)
    theta = [0.8, 6.0, 7.5, 9.0]
    old = mod.SIGMA2_IN_TRAIN_2D
    try:
        mod.SIGMA2_IN_TRAIN_2D = s2
        gp = cg.gaussian_process(y, x, kernel='RBF2D', y_err=e)
        gp.nugget = 0.05
        g = gp.compute_log_likelihood_gradient(theta)
    finally:
        mod.SIGMA2_IN_TRAIN_2D = old
    ref = orc.log_likelihood_gradient(y, x, 'RBF2D', theta, 0.05, y_err=e, y_mean=0., sigma2_in_train=s2)
    np.testing.assert_allclose(g, ref[:4], rtol=1e-8, atol=1e-8)


@pytest.mark.gpu
def test_gpu_gradient_nobject_and_lbfgs_fit(cg):
    g = load_golden('predict_nobject_1d')
    n = int(g['nobj'])
    xs, ys, es = ([g['%s_%d' % (k, i)] for i in range(n)] for k in 'xye')
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    h = list(g['hyp'])
    got = gp.compute_log_likelihood_gradient(h)                      # ragged 31 / 44 / 25: one batched launch sequence
    ref = sum(orc.log_likelihood_gradient(ys[i], xs[i], 'RBF1D', h, 0., y_err=es[i], y_mean=0.) for i in range(n))
    np.testing.assert_allclose(got, ref[:2], rtol=1e-8, atol=1e-8)
    ll_batched = gp.log_likelihood.copy()
    gp.compute_log_likelihood(h, svd_method=False)
    assert np.array_equal(ll_batched, gp.log_likelihood)
    # every object alone gives the same numbers bit for bit
    alone = np.zeros(2)
    for i in range(n):
        g1 = cg.gaussian_process(ys[i], xs[i], kernel='RBF1D', y_err=es[i])
        alone = alone + g1.compute_log_likelihood_gradient(h)
    np.testing.assert_allclose(got, alone, rtol=1e-13)
    # a larger, 2D batch (sizes 150 .. 400) and the joint L-BFGS-B fit of shared hyperparameters
    rng = np.random.default_rng(9)
    xs2, ys2, es2 = [], [], []
    for k in (150, 260, 400, 333):
        x = rng.uniform(0., 40., (k, 2))
        e = rng.uniform(0.05, 0.1, k)
        xs2.append(x); es2.append(e)
        ys2.append(np.sin(x[:, 0] / 7.) * np.cos(x[:, 1] / 9.) + e * rng.standard_normal(k))
    g2 = cg.gaussian_process_nobject(ys2, xs2, kernel='RBF2D', y_err=es2)
    h2 = [0.9, 7., 8., 5.]
    got2 = g2.compute_log_likelihood_gradient(h2)
    ref2 = sum(orc.log_likelihood_gradient(ys2[i], xs2[i], 'RBF2D', h2, 0., y_err=es2[i], y_mean=0.) for i in range(4))
    np.testing.assert_allclose(got2, ref2[:4], rtol=1e-8, atol=1e-8)
    gg = load_golden('fit_1d_n60')
    gp = cg.gaussian_process(gg['y'], gg['x'], kernel='RBF1D', y_err=gg['e'])
    gp.find_hyperparameters(hyperparameter_guess=list(gg['guess']), svd_method=False, method='L-BFGS-B')
    np.testing.assert_allclose(gp.hyperparameters, gg['fit_hyp'], rtol=2e-3)     # the reference's fmin optimum
    assert gp.fit_result.nfev < 40


@pytest.mark.gpu
def test_gpu_gradient_large_vs_central_differences(cg):
    """N = 2048 (beyond what the dense oracle gradient does in seconds): central differences of the
    device likelihood itself."""
    rng = np.random.default_rng(5)
    n = 2048
    x = np.sort(rng.uniform(0., 400., n))
    e = np.full(n, 0.1)
    y = np.sin(x / 9.) + 0.1 * rng.standard_normal(n)
    gp = cg.gaussian_process(y, x, kernel='RBF1D', y_err=e)
    theta = np.array([1.1, 3.0])
    g = gp.compute_log_likelihood_gradient(theta)

    def ll(t):
        gp.compute_log_likelihood(t, svd_method=False)
        return gp.log_likelihood[0]

    np.testing.assert_allclose(g, _fd(ll, theta, eps=1e-5), rtol=1e-5, atol=1e-4)
