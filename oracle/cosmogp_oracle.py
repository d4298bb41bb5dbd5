"""CPU oracle for the cosmogp GP hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference algorithm
(PFLeget/cosmogp, Python 2) for the path named in BASELINE.json: RBF covariance
assembly (cosmogp/kernel.py) + Cholesky inverse/logdet (cosmogp/inv_matrix.py)
as driven by cosmogp/Gaussian_process.py (log-likelihood, prediction,
predictive covariance) and its two edge callers (mean.py, pull.py).

Rules (see DESIGN.md "Oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this file; the product package cosmogp_amd never does;
  * it is pinned: tests/golden/*.npz were produced by the real reference
    (loaded in memory with its Python-2 print statements fixed, see
    tests/golden/make_golden.py) and tests/test_oracle_golden.py checks every
    function below against them;
  * the arithmetic that lives in third-party code on the reference side
    (scipy.linalg -> LAPACK dpotrf/dgetri/dgesdd; scipy.spatial.distance
    pdist/cdist 'mahalanobis', unpinned in requirements.txt:1-3) is called
    through SciPy here too, or restated from its published algorithm where
    noted; goldens were made with numpy 2.2.6 / scipy 1.15.3.

Every function cites the reference file:line it follows.  The code is written
as plain functions over lists of per-object arrays (the reference uses one
class with per-object loops), so nothing here is a copy of reference source.
"""
import numpy as np
from scipy import linalg as _la
from scipy import interpolate as _interp
from scipy.optimize import fmin as _fmin


# --------------------------------------------------------------------------
# kernel.py
# --------------------------------------------------------------------------
def init_rbf(x, y):
    """First guess [sigma, L] -- cosmogp/kernel.py:6-22.

    Quirk kept: the per-object scalars are overwritten inside the loop
    (kernel.py:14-17), so only the LAST object's min/max/std count, while
    number_point stays a per-object vector.
    """
    npts = np.array([float(len(xi)) for xi in x])
    lo = np.min(x[-1])
    hi = np.max(x[-1])
    sig = np.std(y[-1])
    d = np.mean(np.sqrt((hi - lo) ** 2 / npts))
    span = np.mean(hi - lo)
    return np.mean(sig), np.mean([d, span])


def diag_noise(n, y_err, floor, nugget):
    """Diagonal term of the train-train matrix -- kernel.py:74-75 / :150-151:
    (y_err**2 + floor**2 + nugget**2), evaluated left to right."""
    if y_err is None:
        y_err = np.zeros(n)
    return np.asarray(y_err, dtype=np.float64) ** 2 + floor ** 2 + nugget ** 2


def rbf_kernel_1d(x, hyperparameter, new_x=None, nugget=0., floor=0., y_err=None):
    """1D squared-exponential covariance -- cosmogp/kernel.py:25-77.

    Train-train (new_x None): (N,N) with the diagonal noise added.
    Cross: (M,N), rows = new points, no diagonal term (kernel.py:64-69).
    Operation order of kernel.py:71-72 is kept: d=x-x2[:,None];
    h0^2 * exp(-0.5 * ((d*d)/h1^2)).
    """
    x = np.asarray(x, dtype=np.float64)
    amp = hyperparameter[0] ** 2
    ell2 = hyperparameter[1] ** 2
    other = x if new_x is None else np.asarray(new_x, dtype=np.float64)
    d = x[None, :] - other[:, None]
    cov = amp * np.exp(-0.5 * ((d * d) / ell2))
    if new_x is None:
        idx = np.arange(len(x))
        cov[idx, idx] += diag_noise(len(x), y_err, floor, nugget)
    return cov


def inv_l_2d(hyperparameter):
    """Inverse of L=[[h1^2,h3],[h3,h2^2]] -- kernel.py:127-130 (same op order:
    build the adjugate, then scale by 1/det)."""
    h = hyperparameter
    m = np.array(([h[2] ** 2, -h[3]], [-h[3], h[1] ** 2]), dtype=np.float64)
    m *= 1. / (((h[1] ** 2) * (h[2] ** 2)) - h[3] ** 2)
    return m


def _mahalanobis_sq_then_sqrt(u, v, vi):
    """scipy.spatial.distance 'mahalanobis' (C kernel mahalanobis_distance in
    scipy/spatial/src/distance_impl.h; third-party, restated from the published
    source): d=u-v; t_i=sum_j d_j*VI[i,j]; s=sum_i d_i*t_i; sqrt(s).
    u: (P,2), v: (Q,2) -> (P,Q)."""
    d0 = u[:, None, 0] - v[None, :, 0]
    d1 = u[:, None, 1] - v[None, :, 1]
    t0 = d0 * vi[0, 0] + d1 * vi[0, 1]
    t1 = d0 * vi[1, 0] + d1 * vi[1, 1]
    s = d0 * t0 + d1 * t1
    return np.sqrt(s)


def rbf_kernel_2d(x, hyperparameter, new_x=None, nugget=0., floor=0., y_err=None,
                  sigma2_in_train=False):
    """Anisotropic 2D squared-exponential -- cosmogp/kernel.py:80-155.

    Cross (kernel.py:137-142): h0^2*exp(-0.5*cdist(x,new_x,VI)^2).T -> (M,N).
    Train-train (kernel.py:143-151): squareform(exp(-0.5*pdist^2)), diagonal
    set to 1, plus diagonal noise -- WITHOUT the h0^2 amplitude (reference bug,
    SURVEY 8a row a2).  sigma2_in_train=True gives the corrected form.
    The reference's two debug prints (kernel.py:138,:153) are not reproduced.
    """
    x = np.asarray(x, dtype=np.float64)
    vi = inv_l_2d(hyperparameter)
    amp = hyperparameter[0] ** 2
    if new_x is not None:
        new_x = np.asarray(new_x, dtype=np.float64)
        dist = _mahalanobis_sq_then_sqrt(x, new_x, vi)
        return (amp * np.exp(-0.5 * dist ** 2)).T.copy()
    dist = _mahalanobis_sq_then_sqrt(x, x, vi)
    cov = np.exp(-0.5 * dist ** 2)
    if sigma2_in_train:
        cov = amp * cov
        np.fill_diagonal(cov, amp)
    else:
        np.fill_diagonal(cov, 1.)
    idx = np.arange(len(x))
    cov[idx, idx] += diag_noise(len(x), y_err, floor, nugget)
    return cov


# --------------------------------------------------------------------------
# inv_matrix.py
# --------------------------------------------------------------------------
def cholesky_inverse(matrix, return_logdet=False):
    """Explicit inverse through the lower Cholesky factor --
    cosmogp/inv_matrix.py:21-31: L=chol(K); iL=inv(L); iL.T@iL; logdet=sum 2 log diag L.
    Raises numpy.linalg.LinAlgError when K is not positive definite."""
    low = _la.cholesky(matrix, lower=True)
    ilow = _la.inv(low)
    inv = np.dot(ilow.T, ilow)
    if return_logdet:
        return inv, np.sum(2. * np.log(np.diag(low)))
    return inv


def svd_inverse(matrix, return_logdet=False):
    """SVD pseudo-inverse, modes with s<=1e-15 dropped -- inv_matrix.py:4-18
    (the 'Pseudo-inverse decomposition' print of :8-9 is not reproduced)."""
    u, s, vt = _la.svd(matrix)
    keep = s > 10 ** -15
    inv = np.dot(vt.T[:, keep], np.dot(np.diag(1. / s[keep]), u.T[keep]))
    if return_logdet:
        return inv, np.sum(np.log(s[keep]))
    return inv


# --------------------------------------------------------------------------
# mean.py
# --------------------------------------------------------------------------
def return_mean(y, x, new_x=None, mean_y=None, mean_xaxis=None, diff=None):
    """Prior mean on the data grid or on new_x -- cosmogp/mean.py:70-104.
    1D (x[0] is np.float64, mean.py:81): cubic InterpolatedUnivariateSpline
    (mean.py:28-31); 2D: bisplrep(task=1)+bisplev per point (mean.py:58-65).
    diff defaults to mean(y - mean_shape) (mean.py:88-89)."""
    def shape_on(grid):
        if mean_y is None:
            return 0
        if type(x[0]) is np.float64:
            return _interp.InterpolatedUnivariateSpline(mean_xaxis, mean_y)(grid)
        tck = _interp.bisplrep(mean_xaxis[:, 0], mean_xaxis[:, 1], mean_y, task=1)
        return np.array([_interp.bisplev(g[0], g[1], tck) for g in grid])

    if mean_y is not None:
        assert mean_xaxis is not None
        assert len(mean_y) == len(mean_xaxis)
    base = shape_on(x)
    if diff is None:
        diff = np.mean(y - base)
    y0 = base + diff
    if new_x is None:
        return y0
    if mean_y is None:
        return y0                      # mean.py:100-101
    return shape_on(new_x) + diff


# --------------------------------------------------------------------------
# Gaussian_process.py
# --------------------------------------------------------------------------
_KERNELS = {'RBF1D': rbf_kernel_1d, 'RBF2D': rbf_kernel_2d}


def log_likelihood_gp(y, x_axis, kernel, hyperparameter, nugget, y_err=None,
                      y_mean=None, svd_method=True):
    """Log marginal likelihood of one object -- Gaussian_process.py:13-75.
    Returns an ndarray of shape (1,) like the reference (:68-70)."""
    kern = _KERNELS[kernel] if isinstance(kernel, str) else kernel
    if y_mean is None:
        y_mean = np.zeros_like(y)
    n = len(x_axis)
    kmat = kern(x_axis, hyperparameter, nugget=nugget, y_err=y_err)
    inv, logdet = (svd_inverse if svd_method else cholesky_inverse)(kmat, return_logdet=True)
    col = (y.reshape(n, 1) - (np.ones(n) * y_mean).reshape(n, 1))
    ll = -0.5 * np.dot((y - y_mean), np.dot(inv, col))
    ll += -(n / 2.) * np.log(2 * np.pi)
    ll -= 0.5 * logdet
    return ll


def object_means(y, x, mean_y=None, mean_xaxis=None, diff=None, substract_mean=False):
    """y0 per object as built by Gaussian_process.__init__ (:175-186)."""
    n_obj = len(y)
    if diff is None:
        diff = [None] * n_obj
    if substract_mean or mean_y is not None:
        return [return_mean(y[i], x[i], mean_y=mean_y, mean_xaxis=mean_xaxis, diff=diff[i])
                for i in range(n_obj)]
    return np.zeros(n_obj)


def total_log_likelihood(y, x, kernel, hyperparameter, nugget, y_err, y0, svd_method=True):
    """Sum over objects -- Gaussian_process.compute_log_likelihood (:191-213)
    with fit_nugget False."""
    tot = 0
    for i in range(len(y)):
        tot = tot + log_likelihood_gp(y[i], x[i], kernel, hyperparameter, nugget,
                                      y_err=y_err[i], y_mean=y0[i], svd_method=svd_method)
    return tot


def find_hyperparameters(y, x, kernel, guess, y_err, y0, nugget=0., fit_nugget=False,
                         svd_method=True):
    """Nelder-Mead on -loglik -- Gaussian_process.find_hyperparameters (:216-253).
    Returns (|hyper|, |nugget|)."""
    x0 = list(guess) + ([1.] if fit_nugget else [])

    def cost(p):
        if fit_nugget:
            hyp, ng = p[:-1], p[-1]
        else:
            hyp, ng = p, nugget
        return -total_log_likelihood(y, x, kernel, hyp, ng, y_err, y0, svd_method)[0]

    best = _fmin(cost, x0, disp=False)
    nh = len(guess)
    hyp = np.array([np.sqrt(best[i] ** 2) for i in range(nh)])
    ng = np.sqrt(best[-1] ** 2) if fit_nugget else nugget
    return hyp, ng


def predict(y, x, kernel, hyperparameter, nugget, y_err, new_binning=None,
            mean_y=None, mean_xaxis=None, diff=None, substract_mean=False,
            svd_method=True, cov=True, fix_stale_index=True):
    """Prediction + predictive covariance for a list of objects --
    Gaussian_process.get_prediction (:270-338) and get_covariance_matrix (:340-361).

    Returns dict(prediction=[(M,)], covariance=[(M,M)] or None,
    kernel_matrix=[(N,N)], inv_kernel_matrix=[(N,N)]).
    fix_stale_index: the reference picks new_binning[i] with the stale loop
    index i=N_sn-1 (:330); True uses the object's own grid (identical for one
    object or for an explicit shared grid)."""
    kern = _KERNELS[kernel]
    n_obj = len(y)
    if diff is None:
        diff = [None] * n_obj
    use_mean = substract_mean or mean_y is not None
    y0 = object_means(y, x, mean_y, mean_xaxis, diff, substract_mean)
    same = new_binning is None
    out = dict(prediction=[], covariance=[] if cov else None,
               kernel_matrix=[], inv_kernel_matrix=[])
    for sn in range(n_obj):
        kmat = kern(x[sn], hyperparameter, nugget=nugget, y_err=y_err[sn])   # :266
        if use_mean:
            new_y0 = y0[sn] if same else return_mean(y[sn], x[sn], new_x=new_binning,
                                                     mean_y=mean_y, mean_xaxis=mean_xaxis,
                                                     diff=diff[sn])
        else:
            new_y0 = 0.
        inv = svd_inverse(kmat) if svd_method else cholesky_inverse(kmat)    # :319-322
        if same:
            grid = x[sn] if fix_stale_index else x[n_obj - 1]
        else:
            grid = new_binning
        h = kern(x[sn], hyperparameter, new_x=grid)                         # :332
        resid = (y[sn] - y0[sn]).reshape(len(y[sn]), 1)
        pred = np.dot(h, np.dot(inv, resid)).T[0] + new_y0                  # :334-335
        out['prediction'].append(pred)
        out['kernel_matrix'].append(kmat)
        out['inv_kernel_matrix'].append(inv)
        if cov:
            grid_c = x[sn] if same else new_binning                        # :351-354
            hc = kern(x[sn], hyperparameter, new_x=grid_c)
            kss = kern(grid_c, hyperparameter, nugget=nugget)               # :357
            out['covariance'].append(-np.dot(hc, np.dot(inv, hc.T)) + kss)  # :359-361
    return out


# --------------------------------------------------------------------------
# pull.py
# --------------------------------------------------------------------------
def compute_pull(y, x, hyperparameters, nugget=0., y_err=None, y_mean=None,
                 x_axis_mean=None, kernel='RBF1D', diff=None, svd_method=True,
                 substract_mean=False):
    """Leave-one-out pulls -- cosmogp/pull.py:43-102.  For each object and each
    point t: refit on the other N-1 points, predict on the full grid, read
    point t.  Returns (pull list, residual list, per-object predictions)."""
    pulls, resids, preds = [], [], []
    for sn in range(len(y)):
        n = len(y[sn])
        yerr = np.zeros_like(y[sn]) if y_err is None else y_err[sn]
        dd = None if diff is None else [diff[sn]]
        pred = np.zeros(n)
        var = np.zeros(n)
        for t in range(n):
            keep = np.ones(n, dtype=bool)
            keep[t] = False
            if y_mean is None and substract_mean:                          # pull.py:71-73
                y_mean = np.ones_like(y[sn]) * np.mean(y[sn])
                x_axis_mean = x[sn]
            res = predict([y[sn][keep]], [x[sn][keep]], kernel, hyperparameters, nugget,
                          [yerr[keep]], new_binning=x[sn], mean_y=y_mean,
                          mean_xaxis=x_axis_mean, diff=dd, substract_mean=substract_mean,
                          svd_method=svd_method, cov=True)
            pred[t] = res['prediction'][0][t]
            var[t] = abs(res['covariance'][0][t, t])
        p = (pred - y[sn]) / np.sqrt(yerr ** 2 + var + nugget ** 2)
        pulls.extend(p.tolist())
        resids.extend((pred - y[sn]).tolist())
        preds.append(pred)
    return np.array(pulls), np.array(resids), preds


# --------------------------------------------------------------------------
# Solve-based forms of the same quantities (used as the large-N comparator:
# the explicit-inverse route above is O(3 N^3) and carries cond(K)*eps error)
# --------------------------------------------------------------------------
def fit_predict_cho(kmat, resid, h, kss_diag):
    """Textbook cho_solve GP (the identity the reference author checks against,
    tests/old/compare_cosmogp_george.py:77-99): returns (mu, var_diag, loglik)."""
    n = kmat.shape[0]
    c = _la.cho_factor(kmat, lower=True)
    alpha = _la.cho_solve(c, resid)
    mu = h @ alpha
    v = _la.solve_triangular(c[0], h.T, lower=True)
    var = kss_diag - np.sum(v * v, axis=0)
    logdet = 2. * np.sum(np.log(np.diag(c[0])))
    ll = -0.5 * resid @ alpha - 0.5 * n * np.log(2 * np.pi) - 0.5 * logdet
    return mu, var, ll


# --------------------------------------------------------------------------
# Analytic gradient of the log marginal likelihood (checker for cgp_fit_grad).
# The reference has no gradient (it searches with scipy.optimize.fmin,
# Gaussian_process.py:247), so there is no golden vector for it: this
# restatement is pinned by central differences of log_likelihood_gp above,
# which IS pinned by the reference's goldens (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------
def log_likelihood_gradient(y, x_axis, kernel, hyperparameter, nugget, y_err=None, y_mean=None,
                            sigma2_in_train=False):
    """d logL / d [hyperparameter..., nugget] = 1/2 tr((a a' - K^-1) dK/dtheta), a = K^-1 (y - y_mean),
    with dK/dtheta written out per hyperparameter of kernel.py:72-75 (1D) / :127-151 (2D)."""
    x = np.asarray(x_axis, dtype=np.float64)
    n = len(x)
    h = np.asarray(hyperparameter, dtype=np.float64)
    if y_mean is None:
        y_mean = np.zeros(n)
    r = np.asarray(y, dtype=np.float64) - np.ones(n) * y_mean
    if kernel == 'RBF1D':
        kmat = rbf_kernel_1d(x, h, nugget=nugget, y_err=y_err)
        d = x[None, :] - x[:, None]
        e = np.exp(-0.5 * d * d / h[1] ** 2)
        dks = [2. * h[0] * e, h[0] ** 2 * e * d * d / h[1] ** 3]
    else:
        kmat = rbf_kernel_2d(x, h, nugget=nugget, y_err=y_err, sigma2_in_train=sigma2_in_train)
        vi = inv_l_2d(h)
        d = x[None, :, :] - x[:, None, :]                       # (n, n, 2)
        q = np.einsum('ija,ab,ijb->ij', d, vi, d)
        amp = h[0] ** 2 if sigma2_in_train else 1.
        e = np.exp(-0.5 * q)
        dks = [2. * h[0] * e if sigma2_in_train else np.zeros((n, n))]
        for dl in (np.array([[2. * h[1], 0.], [0., 0.]]), np.array([[0., 0.], [0., 2. * h[2]]]),
                   np.array([[0., 1.], [1., 0.]])):
            dvi = -vi.dot(dl).dot(vi)
            dks.append(amp * e * (-0.5) * np.einsum('ija,ab,ijb->ij', d, dvi, d))
    dks.append(2. * nugget * np.eye(n))
    c = _la.cho_factor(kmat, lower=True)
    alpha = _la.cho_solve(c, r)
    m = np.outer(alpha, alpha) - _la.cho_solve(c, np.eye(n))
    return np.array([0.5 * np.sum(m * dk) for dk in dks])
