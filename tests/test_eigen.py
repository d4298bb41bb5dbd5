"""Device eigensolver route = the reference's svd_method=True algorithm (cosmogp/inv_matrix.py:4-18,
SURVEY.md 8 row a5 / 8(f) rank 3), through the C ABI: cgp_svd_inverse (operator level) and cgp_fit_svd
(fused).  Tolerances: for positive definite K two backward-stable decompositions agree to ~cond(K) eps.
For numerically singular K parity is UNPINNED: the reference keeps or drops a mode by comparing a singular
value that is pure rounding noise with 1e-15 (on the fixture below LAPACK's dgesdd returns a 3.4e-15 plateau
for the 44 null modes -- all kept -- while scipy.linalg.svdvals returns 1e-17..1e-18 for the same matrix), so
only implementation-independent properties are checked there."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import cosmogp_oracle as orc

pytestmark = pytest.mark.gpu


import sys


@pytest.fixture(scope='module')
def cg():
    import cosmogp_amd
    cosmogp_amd._lib.default_device()
    return cosmogp_amd


@pytest.fixture
def eigen_route(cg):
    """svd_method=True literally = the device eigensolver (the default 'auto' would serve these noisy
    objects by the Cholesky route)."""
    gpm = sys.modules['cosmogp_amd.gaussian_process']
    old = gpm.SVD_VIA_CHOLESKY_WHEN_PD
    gpm.SVD_VIA_CHOLESKY_WHEN_PD = False
    yield gpm
    gpm.SVD_VIA_CHOLESKY_WHEN_PD = old


def test_svd_inverse_golden_pd(cg):
    g = load_golden('inverse_n96')
    inv, ld = cg.svd_inverse(g['K'], return_logdet=True)
    np.testing.assert_allclose(inv, g['svd_inv'], rtol=0, atol=1e-10 * np.abs(g['svd_inv']).max())
    assert abs(ld - float(g['svd_logdet'])) <= 1e-11 * abs(ld)
    np.testing.assert_allclose(inv, inv.T, rtol=0, atol=1e-12 * np.abs(inv).max())


def test_svd_inverse_singular_properties(cg):
    g = load_golden('inverse_n96')
    K = g['K_singular']
    inv, ld = cg.svd_inverse(K, return_logdet=True)          # never raises (the Cholesky route does)
    assert np.all(np.isfinite(inv)) and np.isfinite(ld)
    np.testing.assert_allclose(inv, inv.T, rtol=0, atol=1e-6 * np.abs(inv).max())
    # (K K^+ K = K cannot be checked in floating point here: |K^+| ~ 1e12 makes the product itself lose 4 digits.)
    # With a threshold above the noise floor the answer is the well-defined truncated pseudo-inverse:
    dev = cg._lib.default_device()
    inv_t, ld_t, ndrop = dev.svd_inverse(K, thresh=1e-8)
    w, v = np.linalg.eigh(K)
    keep = np.abs(w) > 1e-8
    want = (v[:, keep] / w[keep]) @ v[:, keep].T
    assert ndrop == np.sum(~keep)
    np.testing.assert_allclose(inv_t, want, rtol=0, atol=1e-5 * np.abs(want).max())    # cond of the kept part 4e9
    assert abs(ld_t - np.sum(np.log(np.abs(w[keep])))) <= 1e-5
    np.testing.assert_allclose(K @ inv_t @ K, K, rtol=0, atol=1e-7)


@pytest.mark.parametrize('n', [1, 2, 5, 37, 64, 65, 128, 200, 515])
def test_svd_inverse_random_spd(cg, n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    q, _ = np.linalg.qr(a)
    lam = np.logspace(0, -6, n)
    K = (q * lam) @ q.T
    K = 0.5 * (K + K.T)
    inv, ld = cg.svd_inverse(K, return_logdet=True)
    ref, ldr = orc.svd_inverse(K, return_logdet=True)
    np.testing.assert_allclose(inv, ref, rtol=0, atol=5e-9 * np.abs(ref).max())       # ~20 cond eps, cond = 1e6
    assert abs(ld - ldr) <= 1e-11 * abs(ldr)
    np.testing.assert_allclose(K @ inv, np.eye(n), rtol=0, atol=1e-8)


def test_svd_inverse_indefinite(cg):
    """s = |lambda|: negative eigenvalues keep their sign in the inverse, logdet uses |lambda|."""
    rng = np.random.default_rng(3)
    n = 150
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.linspace(1., 3., n // 2), -np.linspace(0.5, 2., n - n // 2)])
    K = (q * lam) @ q.T
    K = 0.5 * (K + K.T)
    inv, ld = cg.svd_inverse(K, return_logdet=True)
    ref, ldr = orc.svd_inverse(K, return_logdet=True)
    np.testing.assert_allclose(inv, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    assert abs(ld - ldr) <= 1e-11 * abs(ldr)


def test_fit_svd_vs_oracle_2d(cg, eigen_route):
    rng = np.random.default_rng(8)
    n, m = 300, 70
    x = rng.uniform(0., 30., (n, 2))
    e = rng.uniform(0.05, 0.15, n)
    y = np.sin(x[:, 0] / 6.) * np.cos(x[:, 1] / 7.) + e * rng.standard_normal(n)
    xn = rng.uniform(0., 30., (m, 2))
    hyp = [0.8, 6.0, 7.5, 9.0]
    gp = cg.gaussian_process(y, x, kernel='RBF2D', y_err=e)
    gp.hyperparameters = hyp
    gp.nugget = 0.03
    gp.get_prediction(new_binning=xn)                              # reference default: svd_method=True
    want = orc.predict([y], [x], 'RBF2D', hyp, 0.03, [e], new_binning=xn, svd_method=True)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.covariance_matrix[0], want['covariance'][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.inv_kernel_matrix[0], want['inv_kernel_matrix'][0], rtol=0,
                               atol=1e-9 * np.abs(want['inv_kernel_matrix'][0]).max())
    gp.compute_log_likelihood(hyp)                                 # svd_method=True default
    ref = orc.log_likelihood_gp(y, x, 'RBF2D', hyp, 0.03, y_err=e, y_mean=0., svd_method=True)
    np.testing.assert_allclose(gp.log_likelihood, ref, rtol=1e-11)
    # and the two routes agree with each other on a positive definite K
    ll_svd = gp.log_likelihood.copy()
    gp.compute_log_likelihood(hyp, svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, ll_svd, rtol=1e-11)


def test_fit_svd_default_fit_reaches_reference_optimum(cg, eigen_route):
    gg = load_golden('fit_1d_n60')
    gp = cg.gaussian_process(gg['y'], gg['x'], kernel='RBF1D', y_err=gg['e'])
    gp.find_hyperparameters(hyperparameter_guess=list(gg['guess']))          # all defaults, like a reference user
    np.testing.assert_allclose(gp.hyperparameters, gg['fit_hyp'], rtol=1e-4)


def test_eigen_n1024_accuracy(cg):
    rng = np.random.default_rng(21)
    n = 1024
    x = np.sort(rng.uniform(0., 100., n))
    d = x[:, None] - x[None, :]
    K = np.exp(-0.5 * d * d / 4.) + 1e-2 * np.eye(n)
    inv, ld = cg.svd_inverse(K, return_logdet=True)
    sign, ldr = np.linalg.slogdet(K)
    assert abs(ld - ldr) <= 1e-10 * abs(ldr)
    np.testing.assert_allclose(K @ inv, np.eye(n), rtol=0, atol=1e-9)
    assert cg._lib.default_device() is not None


def test_auto_route_selection(cg):
    """Default 'auto': noisy objects (no mode can be dropped) are served by the Cholesky route, noise-free
    ones by the eigensolver; both without the caller changing svd_method."""
    rng = np.random.default_rng(4)
    n = 150
    x = np.sort(rng.uniform(0., 20., n))
    y = np.sin(x / 2.)
    calls = []
    dev = cg._lib.Device()
    real = dev.fit_svd
    dev.fit_svd = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    gp = cg.gaussian_process(y, x, kernel='RBF1D', y_err=np.full(n, 0.1), device=dev)
    gp.hyperparameters = [1., 2.]
    gp.get_prediction(new_binning=np.linspace(0., 20., 40))          # svd_method=True by default
    gp.compute_log_likelihood([1., 2.])
    assert calls == []
    want = orc.predict([y], [x], 'RBF1D', [1., 2.], 0., [np.full(n, 0.1)], new_binning=np.linspace(0., 20., 40),
                       svd_method=True)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.covariance_matrix[0], want['covariance'][0], rtol=0, atol=1e-9)
    g0 = cg.gaussian_process(y, x, kernel='RBF1D', device=dev)        # y_err = None: K is numerically singular
    g0.hyperparameters = [1., 2.]
    g0.get_prediction(new_binning=np.linspace(0., 20., 40))
    assert len(calls) == 1 and np.all(np.isfinite(g0.Prediction[0]))
    with pytest.raises(np.linalg.LinAlgError):
        g0.get_prediction(new_binning=np.linspace(0., 20., 40), svd_method=False)


def test_low_rank_pre_stage_equals_full_sweeps(cg, monkeypatch):
    """Noise-free kernel matrices are numerically of low rank; from 256 points on the eigen route first runs a pivoted Cholesky
    (rank r), orthonormalises its factor (shifted Cholesky-QR) and diagonalises the r x r Rayleigh-Ritz matrix instead of sweeping
    the N x N one (csrc/jacobi.hpp).  With a threshold above the rounding noise (1e-9 |K|: the reference's absolute 1e-15 sits below
    it for a matrix of this scale, where K^+ is not reproducible by ANY method) the two must agree: same kept / dropped split,
    same log-determinant, same K^+ y for a y in the range of K."""
    from cosmogp_amd import _lib
    n = 700
    rng = np.random.default_rng(3)
    x = np.sort(rng.uniform(0., 70., n))
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 4.)
    y = K @ (np.sin(0.3 * x) + 0.5 * np.cos(0.11 * x))
    out = {}
    for mode in ('lowrank', 'full'):
        if mode == 'full':
            monkeypatch.setenv('CGP_JAC_NO_LOWRANK', '1')
        else:
            monkeypatch.delenv('CGP_JAC_NO_LOWRANK', raising=False)
        dev = _lib.Device(0)
        out[mode] = dev.svd_inverse(K, thresh=1e-9)
        dev.close()
    (il, ll, dl), (if_, lf, df) = out['lowrank'], out['full']
    w = np.linalg.eigvalsh(K)
    assert dl == df == int(np.sum(w <= 1e-9)) and dl > n // 2            # really rank deficient, same split
    # (the smallest kept eigenvalues, ~1e-9, carry absolute errors of ~eps |K| = 1e-14 in every method: 1e-5 in their logarithm)
    assert abs(ll - lf) <= 1e-4 and abs(ll - np.sum(np.log(w[w > 1e-9]))) <= 1e-4
    al, af = il @ y, if_ @ y
    assert np.abs(K @ al - y).max() <= 1e-6 * np.abs(y).max()
    assert np.abs(K @ af - y).max() <= 1e-6 * np.abs(y).max()
    # (alpha = K^+ y itself is ill-conditioned at this threshold -- its components along the 1e-9 modes carry eps |y| / 1e-9 --
    # the prediction H alpha is not)
    xs = np.linspace(1., 69., 40)
    H = np.exp(-0.5 * (xs[:, None] - x[None, :]) ** 2 / 4.)
    np.testing.assert_allclose(H @ al, H @ af, rtol=0, atol=1e-6 * np.abs(H @ af).max())
    np.testing.assert_allclose(il, il.T, rtol=0, atol=1e-9 * np.abs(il).max())
    # full rank (noise on the diagonal): the pre-stage gives up and the full sweeps answer, as before
    monkeypatch.delenv('CGP_JAC_NO_LOWRANK', raising=False)
    dev = _lib.Device(0)
    Kn = K[:300, :300] + 1e-2 * np.eye(300)
    inv, ld, nd = dev.svd_inverse(Kn)
    dev.close()
    assert nd == 0 and np.abs(Kn @ inv - np.eye(300)).max() < 1e-9


def test_low_rank_pre_stage_effective_threshold_is_the_documented_one(cg, monkeypatch):
    """Round-3 advisor finding, answered by a stated contract (include/cosmogp_hip.h, cgp_svd_inverse): at the reference's
    ABSOLUTE 1e-15 on a noise-free matrix of scale ~1 the two routes do NOT agree -- the sweeps keep the rounding-noise modes
    between 1e-15 and 8 n eps max K_ii, the pre-stage drops them -- and that is all they differ by: the pre-stage never drops a
    mode above its documented effective threshold, it never keeps one the sweeps drop, and the surplus of the sweeps lies
    entirely below the bound (checked against numpy's eigenvalues of the same matrix)."""
    from cosmogp_amd import _lib
    rng = np.random.default_rng(21)
    n = 400
    x = np.sort(rng.uniform(0., 40., n))
    K = orc.rbf_kernel_1d(x, [1.0, 2.0])                         # scale 1, numerically of rank ~60
    bound = 8. * n * np.finfo(float).eps * K.diagonal().max()
    w = np.abs(np.linalg.eigvalsh(K))
    out = {}
    for mode in ('lowrank', 'full'):
        if mode == 'full':
            monkeypatch.setenv('CGP_JAC_NO_LOWRANK', '1')
        else:
            monkeypatch.delenv('CGP_JAC_NO_LOWRANK', raising=False)
        d = _lib.Device(0)
        inv, ld, nd = d.svd_inverse(K, thresh=1e-15)
        out[mode] = (ld, nd, d.eig_info())
        d.close()
    (ll, dl, info_l), (lf, df, info_f) = out['lowrank'], out['full']
    assert info_l[0] == 1 and info_f[0] == 0
    assert dl >= df                                              # the pre-stage drops at least what the sweeps drop ...
    assert n - dl >= np.sum(w > 16. * bound)                     # ... but nothing clearly above its effective threshold,
    assert n - df >= np.sum(w > 16. * bound)
    assert n - dl <= info_l[1]                                    # and keeps no more modes than the rank it found
    # every extra mode of the sweeps is noise below the bound: it pulls their log-determinant DOWN by at least |log bound| each
    # (up to the O(bound) errors both routes carry on the smallest shared modes: one unit of slack per kept mode near the bound)
    if dl > df:
        near = int(np.sum((w > bound / 64.) & (w < 64. * bound)))
        assert lf - ll <= (dl - df) * np.log(16. * bound) + 5. * (near + 1)


def test_noise_free_prediction_through_the_classes(cg):
    """The reference's default call (svd_method=True) on noise-free data of 400 points, small amplitude (so that the absolute
    1e-15 threshold lies above the rounding noise and the kept / dropped split is a property of the matrix): prediction and
    variance against the oracle's truncated-SVD route."""
    n = 400
    rng = np.random.default_rng(8)
    x = np.sort(rng.uniform(0., 40., n))
    hyp = [0.004, 1.5]
    y = 0.004 * (np.sin(0.7 * x) + 0.3 * np.cos(1.9 * x))
    xn = np.linspace(1., 39., 23)
    gp = cg.gaussian_process(y, x, kernel='RBF1D')
    gp.hyperparameters = np.array(hyp)
    gp.get_prediction(new_binning=xn)                     # svd_method=True by default; no noise: the eigen route
    want = orc.predict([y], [x], 'RBF1D', hyp, 0., [np.zeros(n)], new_binning=xn, svd_method=True)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=1e-6 * hyp[0])
    np.testing.assert_allclose(gp.variance[0], np.diag(want['covariance'][0]), rtol=0, atol=1e-6 * hyp[0] ** 2)


@pytest.mark.parametrize('ell', [3.0, 0.45])
def test_noise_free_prediction_rank_below_and_above_the_cap(cg, ell, monkeypatch):
    """Noise-free, 320 points on 40 units, small amplitude (so that the reference's absolute 1e-15 threshold lies above the rounding
    noise): with length scale 3 the matrix has a few dozen resolved modes and the low-rank pre-stage answers; with 0.45 its
    numerical rank exceeds the cap of N / 2 pivoted-Cholesky steps and the full sweeps must take over from an UNTOUCHED K.  Either
    way the prediction equals the oracle's truncated-SVD route.  (1D on purpose: the reference's 2D train-train matrix carries no
    sigma^2, kernel.py:146, so a 2D matrix always has scale 1 and its 1e-15 modes are rounding noise.)"""
    from cosmogp_amd import _lib
    n = 320
    rng = np.random.default_rng(21)
    x = np.sort(rng.uniform(0., 40., n))
    sig = 0.004
    hyp = [sig, ell]
    y = sig * (np.sin(0.4 * x) + 0.4 * np.cos(1.3 * x))
    xn = np.linspace(1., 39., 21)
    dev = _lib.Device(0)
    gp = cg.gaussian_process(y, x, kernel='RBF1D', device=dev)
    gp.hyperparameters = np.array(hyp)
    gp.get_prediction(new_binning=xn)                      # svd_method=True, no noise: the eigen route
    want = orc.predict([y], [x], 'RBF1D', hyp, 0., [np.zeros(n)], new_binning=xn, svd_method=True)
    # (length scale 0.45: dozens of modes crowd around the threshold, the kept / dropped split of two decompositions differs by
    # a mode or two and the prediction by a few 1e-6 sigma)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=(2e-6 if ell > 1. else 1e-5) * sig)
    dev.close()
    if ell < 1.:
        # the pre-stage gave up: the answer must be bit for bit that of the full sweeps alone
        monkeypatch.setenv('CGP_JAC_NO_LOWRANK', '1')
        dev2 = _lib.Device(0)
        g2 = cg.gaussian_process(y, x, kernel='RBF1D', device=dev2)
        g2.hyperparameters = np.array(hyp)
        g2.get_prediction(new_binning=xn)
        assert np.array_equal(g2.Prediction[0], gp.Prediction[0]) and np.array_equal(g2.variance[0], gp.variance[0])
        dev2.close()


def test_full_rank_shortcut_is_certified_and_equals_the_sweeps(cg, monkeypatch):
    """svd_inverse of a positive definite, well-conditioned matrix (noisy data with the reference's default svd_method=True):
    no singular value can be at or below the threshold -- certified on the device by lambda_min >= 1 / |K^-1|_F -- so the
    pseudo-inverse is the inverse and the eigen route answers from a Cholesky factorisation (cgp_eig_info route 2) instead of
    ~18 sweeps.  Same inverse, log-determinant and dropped count as the sweeps (CGP_EIG_NO_CHOL) and as the oracle."""
    from cosmogp_amd import _lib
    n = 700
    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(0., 70., n))
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 4.) + np.diag(rng.uniform(0.01, 0.04, n))
    out = {}
    for mode in ('shortcut', 'sweeps'):
        if mode == 'sweeps':
            monkeypatch.setenv('CGP_EIG_NO_CHOL', '1')
        else:
            monkeypatch.delenv('CGP_EIG_NO_CHOL', raising=False)
        dev = _lib.Device(0)
        out[mode] = dev.svd_inverse(K) + (dev.eig_info(),)
        dev.close()
    (i2, l2, d2, r2), (i0, l0, d0, r0) = out['shortcut'], out['sweeps']
    assert r2[0] == 2 and r2[2] == 0 and r0[0] == 0 and r0[2] > 3
    assert d2 == d0 == 0
    ref, ldr = orc.svd_inverse(K, return_logdet=True)
    assert abs(l2 - l0) <= 1e-11 * abs(l0) and abs(l2 - ldr) <= 1e-11 * abs(ldr)
    np.testing.assert_allclose(i2, i0, rtol=0, atol=1e-10 * np.abs(i0).max())       # cond ~ 4e3
    np.testing.assert_allclose(i2, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    np.testing.assert_array_equal(i2, i2.T)


def test_full_rank_shortcut_is_refused_when_it_cannot_be_certified(cg):
    """The certificate must fail -- and the sweeps answer as before -- for (a) a positive definite matrix whose condition number
    is beyond what the computed inverse can vouch for, (b) an indefinite matrix (s = |lambda|: the Cholesky factorisation stops),
    (c) a threshold the smallest eigenvalue does not clear by the safety factor: modes may have to go."""
    from cosmogp_amd import _lib
    n = 300
    rng = np.random.default_rng(11)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    dev = _lib.Device(0)

    def sym(lam):
        K = (q * lam) @ q.T
        return 0.5 * (K + K.T)
    # (a) cond 1e12
    K = sym(np.logspace(0, -12, n))
    inv, ld, nd = dev.svd_inverse(K)
    assert dev.eig_info()[0] == 0 and nd == 0
    assert abs(ld - np.sum(np.log(np.logspace(0, -12, n)))) <= 1e-2
    # (b) indefinite
    lam = np.concatenate([np.linspace(1., 3., n // 2), -np.linspace(0.5, 2., n - n // 2)])
    K = sym(lam)
    inv, ld, nd = dev.svd_inverse(K)
    assert dev.eig_info()[0] == 0 and nd == 0
    np.testing.assert_allclose(inv, sym(1. / lam), rtol=0, atol=1e-11 * np.abs(inv).max())
    # (c) well conditioned, but the caller's threshold cuts into the spectrum
    lam = np.logspace(0, -3, n)
    K = sym(lam)
    inv, ld, nd = dev.svd_inverse(K, thresh=1e-2)
    assert dev.eig_info()[0] == 0 and nd == int(np.sum(lam <= 1e-2))
    inv, ld, nd = dev.svd_inverse(K, thresh=1e-7)          # ... and with a threshold far below it the shortcut answers
    assert dev.eig_info()[0] == 2 and nd == 0
    np.testing.assert_allclose(inv, sym(1. / lam), rtol=0, atol=1e-10 * np.abs(inv).max())
    dev.close()


def test_forced_eigen_route_on_noisy_data_through_the_classes(cg, eigen_route):
    """svd_method=True taken literally on noisy data of 600 points (SVD_VIA_CHOLESKY_WHEN_PD off): cgp_fit_svd, whose eigen route
    certifies full rank on the device; prediction, covariance and likelihood against the oracle's SVD route."""
    rng = np.random.default_rng(17)
    n, m = 600, 50
    x = np.sort(rng.uniform(0., 60., n))
    e = rng.uniform(0.05, 0.15, n)
    y = np.sin(x / 3.) + e * rng.standard_normal(n)
    xn = np.linspace(1., 59., m)
    dev = cg._lib.Device(0)
    gp = cg.gaussian_process(y, x, kernel='RBF1D', y_err=e, device=dev)
    gp.hyperparameters = [1.1, 2.5]
    gp.get_prediction(new_binning=xn)
    assert dev.eig_info()[0] == 2
    want = orc.predict([y], [x], 'RBF1D', [1.1, 2.5], 0., [e], new_binning=xn, svd_method=True)
    np.testing.assert_allclose(gp.Prediction[0], want['prediction'][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.covariance_matrix[0], want['covariance'][0], rtol=0, atol=1e-9)
    gp.compute_log_likelihood([1.1, 2.5])
    ref = orc.log_likelihood_gp(y, x, 'RBF1D', [1.1, 2.5], 0., y_err=e, y_mean=0., svd_method=True)
    np.testing.assert_allclose(gp.log_likelihood, ref, rtol=1e-11)
    dev.close()
