"""pytest configuration: markers and shared helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    """Fixtures are plain .npz data (no pickles)."""
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def ulp_diff(a, b):
    """Max distance in units of the last place between two float64 arrays."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2 ** 63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2 ** 63) - ib[ib < 0]
    return int(np.max(np.abs(ia - ib))) if a.size else 0


def oracle_fit_with_bounds(kmat, resid, h, kss_diag, c=2. ** -7):
    """The textbook cho_solve GP of the oracle (oracle.fit_predict_cho: inv_matrix.py:23 + Gaussian_process.py:68-73, :334-361)
    together with COMPUTED error bounds for a comparison at this matrix' conditioning (round 4: the full-size parity tests used
    hand-written tolerances with a cond(K) argument in a comment).  cond_1(K) is estimated from the oracle's own factor
    (LAPACK dpocon), and with u = c * cond_1(K) * eps the componentwise forward-error model of a Cholesky solve gives
        |d alpha| <= u |alpha|   ->   |d mu_t|  <= u * sum_j |H_tj| |alpha_j|
                                      |d var_t| <= u * (|v_t|^2 + kss_t),  v_t = L^-1 h_t
                                      |d loglik| <= u * (1/2 |r|.|alpha| + 1/2 |logdet|)  (+ the n/2 log 2 pi constant is exact)
    Both sides of a comparison (device, oracle) carry such an error, so the assertion uses 2 u.
    The constant: c = 1 is the worst-case model; MEASURED on the full-size configurations (gpurun_out/r04_tests1.log, c = 1):
    C2 (cond 1e4) <= 1.1e-4 of the bound, C3 (7e4) <= 4.7e-5, C4 (3e5) <= 3.9e-5, C5 at N = 16384 (3e4) 8.4e-6 -- rounding
    errors average out instead of adding up --, so the default c = 2^-7 still leaves two orders of margin and is ~100 x
    tighter than the worst case.
    Returns dict(mu, var, ll, cond, bmu, bvar, bll)."""
    from scipy import linalg as la
    n = kmat.shape[0]
    anorm = np.abs(kmat).sum(axis=0).max()
    cf = la.cho_factor(kmat, lower=True)
    rcond, info = la.lapack.dpocon(cf[0], anorm, uplo='L')
    assert info == 0 and rcond > 0.
    cond = 1.0 / rcond
    alpha = la.cho_solve(cf, resid)
    mu = h @ alpha
    v = la.solve_triangular(cf[0], h.T, lower=True)
    v2 = np.sum(v * v, axis=0)
    var = kss_diag - v2
    logdet = 2. * np.sum(np.log(np.diag(cf[0])))
    ll = -0.5 * resid @ alpha - 0.5 * n * np.log(2 * np.pi) - 0.5 * logdet
    u = 2. * c * cond * np.finfo(np.float64).eps
    return dict(mu=mu, var=var, ll=ll, cond=cond, u=u,
                bmu=u * (np.abs(h) @ np.abs(alpha)), bvar=u * (v2 + np.abs(kss_diag)),
                bll=u * (0.5 * np.abs(resid) @ np.abs(alpha) + 0.5 * abs(logdet)))


def assert_within_bounds(o, ll, mu, var, what=''):
    """device (ll, mu, var) against oracle_fit_with_bounds' result `o`; the message carries the measured fraction of each bound"""
    rl = abs(ll - o['ll']) / o['bll']
    rm = float(np.max(np.abs(mu - o['mu']) / o['bmu'])) if len(o['mu']) else 0.
    rv = float(np.max(np.abs(var - o['var']) / o['bvar'])) if len(o['var']) else 0.
    msg = '%s cond_1(K) = %.3g: |d loglik| / bound = %.3g, |d mu| / bound = %.3g, |d var| / bound = %.3g' % (what, o['cond'], rl, rm, rv)
    print(msg)
    assert rl <= 1. and rm <= 1. and rv <= 1., msg
