"""Generate golden vectors from the REAL reference (PFLeget/cosmogp).

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference is Python 2 (print statements at cosmogp/inv_matrix.py:9,
cosmogp/kernel.py:138,:153), so `import cosmogp` raises an ordinary
SyntaxError under Python 3.  This script reads the five module sources as
text from /root/reference, applies lib2to3's `fix_print` IN MEMORY and
exec()s them into a synthetic `cosmogp` package, in the order of
cosmogp/__init__.py:12-25.  Nothing derived from the reference source is
written anywhere: only inputs and numerical outputs (data) are saved.

Every fixture stores its inputs too (legacy seeds are not a stable contract)
plus the numpy / scipy versions that produced it.
"""
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np
import scipy

REF = os.environ.get('COSMOGP_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    warnings.filterwarnings('ignore')
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(['lib2to3.fixes.fix_print'])
    pkg = types.ModuleType('cosmogp')
    pkg.__path__ = []
    sys.modules['cosmogp'] = pkg

    def load(name):
        path = os.path.join(REF, 'cosmogp', name + '.py')
        with open(path) as fh:
            src = fh.read()
        if not src.endswith('\n'):
            src += '\n'
        src3 = str(tool.refactor_string(src, path))
        mod = types.ModuleType('cosmogp.' + name)
        mod.__package__ = 'cosmogp'
        sys.modules['cosmogp.' + name] = mod
        exec(compile(src3, path, 'exec'), mod.__dict__)
        return mod

    inv = load('inv_matrix')
    pkg.svd_inverse = inv.svd_inverse
    pkg.cholesky_inverse = inv.cholesky_inverse
    mean = load('mean')
    pkg.return_mean = mean.return_mean
    kern = load('kernel')
    pkg.init_rbf = kern.init_rbf
    pkg.rbf_kernel_1d = kern.rbf_kernel_1d
    pkg.rbf_kernel_2d = kern.rbf_kernel_2d
    gp = load('Gaussian_process')
    pkg.Gaussian_process = gp.Gaussian_process
    pkg.gaussian_process = gp.gaussian_process
    pkg.gaussian_process_nobject = gp.gaussian_process_nobject
    pkg.log_likelihood_gp = gp.log_likelihood_gp
    pull = load('pull')
    pkg.build_pull = pull.build_pull
    return pkg


@contextlib.contextmanager
def quiet():
    """The reference prints debug text on every 2D kernel call."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def smooth_1d(rng, x, noise):
    y = np.sin(0.9 * x) + 0.4 * np.cos(2.3 * x + 0.3) + 0.05 * x
    return y + noise * rng.standard_normal(len(x))


def smooth_2d(rng, X, noise):
    y = np.sin(0.35 * X[:, 0]) * np.cos(0.27 * X[:, 1]) + 0.02 * X[:, 0]
    return y + noise * rng.standard_normal(len(X))


def save(name, **arrs):
    arrs['numpy_version'] = np.array(np.__version__)
    arrs['scipy_version'] = np.array(scipy.__version__)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrs)
    print('wrote', name)


def main():
    ref = load_reference()
    rng = np.random.default_rng(20261004)

    # ---------------- a1: rbf_kernel_1d ---------------------------------
    for n, m in ((8, 5), (64, 19), (256, 64)):
        x = np.sort(rng.uniform(-12., 42., n))
        xn = np.linspace(-12., 42., m)
        yerr = rng.uniform(0.02, 0.1, n)
        hyp = np.array([0.7, 6.5])
        save('k1d_n%d' % n, x=x, xnew=xn, yerr=yerr, hyp=hyp, nugget=0.03, floor=0.01,
             K_plain=ref.rbf_kernel_1d(x, hyp),
             K_noise=ref.rbf_kernel_1d(x, hyp, nugget=0.03, floor=0.01, y_err=yerr),
             H=ref.rbf_kernel_1d(x, hyp, new_x=xn),
             Kss=ref.rbf_kernel_1d(xn, hyp, nugget=0.03))

    # ---------------- a2: rbf_kernel_2d ---------------------------------
    for n, m in ((8, 5), (64, 19), (200, 36)):
        X = rng.uniform(0., 30., (n, 2))
        Xn = rng.uniform(0., 30., (m, 2))
        yerr = rng.uniform(0.02, 0.1, n)
        hyp = np.array([0.5, 3.1, 2.2, 1.7])
        with quiet():
            save('k2d_n%d' % n, x=X, xnew=Xn, yerr=yerr, hyp=hyp, nugget=0.03, floor=0.01,
                 K_plain=ref.rbf_kernel_2d(X, hyp),
                 K_noise=ref.rbf_kernel_2d(X, hyp, nugget=0.03, floor=0.01, y_err=yerr),
                 H=ref.rbf_kernel_2d(X, hyp, new_x=Xn),
                 Kss=ref.rbf_kernel_2d(Xn, hyp, nugget=0.03))

    # ---------------- a3: init_rbf ---------------------------------------
    xs = [np.sort(rng.uniform(0, 5, k)) for k in (30, 41, 17)]
    ys = [smooth_1d(rng, xi, 0.1) for xi in xs]
    Xs = [rng.uniform(0, 9, (k, 2)) for k in (25, 36)]
    Ys = [smooth_2d(rng, Xi, 0.1) for Xi in Xs]
    pack = {}
    for i, (a, b) in enumerate(zip(xs, ys)):
        pack['x1_%d' % i], pack['y1_%d' % i] = a, b
    for i, (a, b) in enumerate(zip(Xs, Ys)):
        pack['x2_%d' % i], pack['y2_%d' % i] = a, b
    save('init_rbf', n1=3, n2=2, guess_1d=np.array(ref.init_rbf(xs, ys)),
         guess_2d=np.array(ref.init_rbf(Xs, Ys)), **pack)

    # ---------------- a4 / a5: inverses ----------------------------------
    n = 96
    x = np.sort(rng.uniform(0., 20., n))
    yerr = rng.uniform(0.05, 0.2, n)
    hyp = np.array([1.1, 1.6])
    K = ref.rbf_kernel_1d(x, hyp, y_err=yerr)
    ci, cl = ref.cholesky_inverse(K, return_logdet=True)
    with quiet():
        si, sl = ref.svd_inverse(K, return_logdet=True)
    Ksing = ref.rbf_kernel_1d(np.linspace(0, 5, 60), np.array([1., 2.]))   # noise-free
    with quiet():
        ssi, ssl = ref.svd_inverse(Ksing, return_logdet=True)
    try:
        ref.cholesky_inverse(Ksing)
        chol_raises = 0
    except np.linalg.LinAlgError:
        chol_raises = 1
    save('inverse_n96', K=K, chol_inv=ci, chol_logdet=cl, svd_inv=si, svd_logdet=sl,
         K_singular=Ksing, svd_inv_singular=ssi, svd_logdet_singular=ssl,
         chol_raises_on_singular=chol_raises)

    # ---------------- a6 / a7: log-likelihood ----------------------------
    ns = (40, 57, 33)
    xs = [np.sort(rng.uniform(-12., 42., k)) for k in ns]
    es = [rng.uniform(0.02, 0.1, k) for k in ns]
    ys = [smooth_1d(rng, 0.12 * xi, 0.05) for xi in xs]
    hyp = np.array([0.6, 7.])
    gp = ref.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    gp.fit_nugget = False
    gp.nugget = 0.04
    gp.compute_log_likelihood(hyp, svd_method=False)
    ll3_chol = np.array(gp.log_likelihood)
    with quiet():
        gp.compute_log_likelihood(hyp, svd_method=True)
    ll3_svd = np.array(gp.log_likelihood)
    gp.fit_nugget = True
    gp.compute_log_likelihood(np.array([0.6, 7., 0.09]), svd_method=False)
    ll3_fitnug = np.array(gp.log_likelihood)
    g1 = ref.gaussian_process(ys[0], xs[0], kernel='RBF1D', y_err=es[0], substract_mean=True)
    g1.fit_nugget = False
    g1.compute_log_likelihood(hyp, svd_method=False)
    ll1_submean = np.array(g1.log_likelihood)
    pack = {}
    for i in range(3):
        pack['x_%d' % i], pack['y_%d' % i], pack['e_%d' % i] = xs[i], ys[i], es[i]
    save('loglik_1d', nobj=3, hyp=hyp, nugget=0.04, ll3_chol=ll3_chol, ll3_svd=ll3_svd,
         ll3_fitnug=ll3_fitnug, hyp_fitnug=np.array([0.6, 7., 0.09]),
         ll1_submean=ll1_submean, **pack)

    X = rng.uniform(0., 20., (120, 2))
    e = rng.uniform(0.05, 0.1, 120)
    yv = smooth_2d(rng, X, 0.05)
    hyp2 = np.array([0.8, 3.0, 2.5, 1.2])
    g2 = ref.gaussian_process(yv, X, kernel='RBF2D', y_err=e)
    g2.fit_nugget = False
    with quiet():
        g2.compute_log_likelihood(hyp2, svd_method=False)
    save('loglik_2d', x=X, y=yv, e=e, hyp=hyp2, ll_chol=np.array(g2.log_likelihood),
         init_hyp=np.array(ref.gaussian_process(yv, X, kernel='RBF2D', y_err=e).hyperparameters))

    # ---------------- a8: find_hyperparameters (N=60) --------------------
    x = np.linspace(0., 5., 60)
    e = np.full(60, 0.2)
    Kt = ref.rbf_kernel_1d(x, np.array([1., 1.]), y_err=e)
    y = np.linalg.cholesky(Kt) @ rng.standard_normal(60)
    gf = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e)
    init_guess = np.array(gf.hyperparameters)
    gf.find_hyperparameters(hyperparameter_guess=[0.8, 1.4], svd_method=False)
    fit_hyp = np.array(gf.hyperparameters)
    gn = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e)
    gn.find_hyperparameters(hyperparameter_guess=[0.8, 1.4], nugget=True, svd_method=False)
    save('fit_1d_n60', x=x, y=y, e=e, init_guess=init_guess, guess=np.array([0.8, 1.4]),
         fit_hyp=fit_hyp, fit_hyp_nug=np.array(gn.hyperparameters), fit_nugget=gn.nugget)

    # ---------------- a10 / a11: prediction, C1 plumbing (N=256) ----------
    rng1 = np.random.default_rng(1)
    x = np.linspace(0., 5., 256)
    e = np.full(256, 0.2)
    Kt = ref.rbf_kernel_1d(x, np.array([1., 1.]), y_err=e)
    y = np.linalg.cholesky(Kt) @ rng1.standard_normal(256)
    xn = np.linspace(0., 5., 64)
    gp = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e)
    gp.hyperparameters = np.array([1., 1.])
    gp.fit_nugget = False
    gp.compute_log_likelihood(gp.hyperparameters, svd_method=False)
    ll = np.array(gp.log_likelihood)
    gp.get_prediction(new_binning=xn, svd_method=False)
    pred, cov = gp.Prediction[0].copy(), gp.covariance_matrix[0].copy()
    gp.get_prediction(new_binning=None, svd_method=False)
    pred_same, cov_same = gp.Prediction[0].copy(), gp.covariance_matrix[0].copy()
    with quiet():
        gp.get_prediction(new_binning=xn, svd_method=True)
    save('c1_predict_1d_n256', x=x, y=y, e=e, xnew=xn, hyp=np.array([1., 1.]), loglik=ll,
         pred=pred, cov=cov, pred_same=pred_same, cov_diag_same=np.diag(cov_same).copy(),
         pred_svd=gp.Prediction[0].copy(), cov_diag_svd=np.diag(gp.covariance_matrix[0]).copy())

    # prediction with a mean function and a nugget (1D, N=80)
    x = np.sort(rng.uniform(-12., 42., 80))
    e = rng.uniform(0.02, 0.08, 80)
    tm = np.linspace(-15., 45., 61)
    my = 1.5 * np.exp(-0.5 * ((tm - 3.) / 9.) ** 2)
    y = 1.5 * np.exp(-0.5 * ((x - 3.) / 9.) ** 2) + 0.3 + smooth_1d(rng, 0.2 * x, 0.03) * 0.2
    xn = np.linspace(-12., 42., 19)
    gm = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e, Mean_Y=my, Time_mean=tm)
    gm.hyperparameters = np.array([0.25, 6.])
    gm.nugget = 0.02
    gm.get_prediction(new_binning=xn, svd_method=False)
    pm, cm = gm.Prediction[0].copy(), gm.covariance_matrix[0].copy()
    gm.get_prediction(new_binning=None, svd_method=False)
    gd = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e, Mean_Y=my, Time_mean=tm, diff=[0.25])
    gd.hyperparameters = np.array([0.25, 6.])
    gd.get_prediction(new_binning=xn, svd_method=False)
    gs = ref.gaussian_process(y, x, kernel='RBF1D', y_err=e, substract_mean=True)
    gs.hyperparameters = np.array([0.25, 6.])
    gs.get_prediction(new_binning=xn, svd_method=False)
    save('predict_mean_1d_n80', x=x, y=y, e=e, xnew=xn, mean_y=my, mean_x=tm,
         hyp=np.array([0.25, 6.]), nugget=0.02, y0=np.array(gm.y0[0]),
         pred=pm, cov=cm, pred_same=gm.Prediction[0].copy(),
         pred_diff=gd.Prediction[0].copy(), cov_diag_diff=np.diag(gd.covariance_matrix[0]).copy(),
         pred_submean=gs.Prediction[0].copy(), cov_diag_submean=np.diag(gs.covariance_matrix[0]).copy())

    # 2D prediction (N=150, M=36), sigma != 1 so the sigma^2 quirk shows
    X = rng.uniform(0., 20., (150, 2))
    e = rng.uniform(0.05, 0.1, 150)
    yv = smooth_2d(rng, X, 0.05)
    g = np.linspace(1., 19., 6)
    Xn = np.array([[a, b] for a in g for b in g])
    hyp2 = np.array([0.8, 3.0, 2.5, 1.2])
    g2 = ref.gaussian_process(yv, X, kernel='RBF2D', y_err=e)
    g2.hyperparameters = hyp2
    g2.nugget = 0.01
    with quiet():
        g2.get_prediction(new_binning=Xn, svd_method=False)
    save('predict_2d_n150', x=X, y=yv, e=e, xnew=Xn, hyp=hyp2, nugget=0.01,
         pred=g2.Prediction[0].copy(), cov=g2.covariance_matrix[0].copy(),
         inv_kernel_matrix=g2.inv_kernel_matrix[0].copy())

    # n objects, ragged, shared grid and own grids (own grids: reference picks
    # new_binning[N_sn-1] for every H in get_prediction -- stale index -- so
    # only equal-length objects run; both are recorded)
    ns = (31, 44, 25)
    xs = [np.sort(rng.uniform(-12., 42., k)) for k in ns]
    es = [rng.uniform(0.02, 0.1, k) for k in ns]
    ys = [smooth_1d(rng, 0.12 * xi, 0.05) for xi in xs]
    xn = np.linspace(-12., 42., 19)
    gp = ref.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    gp.hyperparameters = np.array([0.6, 7.])
    gp.get_prediction(new_binning=xn, svd_method=False)
    pack = {}
    for i in range(3):
        pack['x_%d' % i], pack['y_%d' % i], pack['e_%d' % i] = xs[i], ys[i], es[i]
        pack['pred_%d' % i] = gp.Prediction[i].copy()
        pack['covdiag_%d' % i] = np.diag(gp.covariance_matrix[i]).copy()
    xs2 = [np.sort(rng.uniform(-12., 42., 30)) for _ in range(3)]
    ys2 = [smooth_1d(rng, 0.12 * xi, 0.05) for xi in xs2]
    es2 = [rng.uniform(0.02, 0.1, 30) for _ in range(3)]
    gq = ref.gaussian_process_nobject(ys2, xs2, kernel='RBF1D', y_err=es2)
    gq.hyperparameters = np.array([0.6, 7.])
    gq.get_prediction(new_binning=None, svd_method=False)
    for i in range(3):
        pack['sx_%d' % i], pack['sy_%d' % i], pack['se_%d' % i] = xs2[i], ys2[i], es2[i]
        pack['spred_stale_%d' % i] = gq.Prediction[i].copy()
        pack['scovdiag_%d' % i] = np.diag(gq.covariance_matrix[i]).copy()
    save('predict_nobject_1d', nobj=3, xnew=xn, hyp=np.array([0.6, 7.]), **pack)

    # ---------------- pull.py (N=20) --------------------------------------
    x = np.linspace(0., 5., 20)
    e = np.full(20, 0.2)
    Kt = ref.rbf_kernel_1d(x, np.array([1., 1.]), y_err=e)
    y = np.linalg.cholesky(Kt) @ rng.standard_normal(20)
    bp = ref.build_pull([y], [x], np.array([1., 1.]), nugget=0.01, y_err=[e])
    bp.compute_pull(svd_method=False)
    bs = ref.build_pull([y], [x], np.array([1., 1.]), nugget=0.01, y_err=[e])
    bs.compute_pull(svd_method=False, substract_mean=True)
    save('pull_1d_n20', x=x, y=y, e=e, hyp=np.array([1., 1.]), nugget=0.01,
         pull=np.array(bp.pull), residual=np.array(bp.residual),
         pull_average=bp.pull_average, pull_std=bp.pull_std,
         pull_submean=np.array(bs.pull), residual_submean=np.array(bs.residual))


def main_round3():
    """Fixtures added in round 3 (own seed: the files of main() are not rewritten): a 2D fit, 2D and ragged n-object
    leave-one-out pulls, and the svd_method=True route on a noise-free (numerically singular) matrix whose spectrum crosses
    the reference's 1e-15 threshold far above the rounding noise of the decomposition."""
    ref = load_reference()
    rng = np.random.default_rng(20261005)

    # ---------------- a8 in 2D: find_hyperparameters (N = 100) -----------
    X = rng.uniform(0., 20., (100, 2))
    e = np.full(100, 0.1)
    hyp_true = np.array([1.0, 3.0, 2.0, 1.5])
    with quiet():
        Kt = ref.rbf_kernel_2d(X, hyp_true, y_err=e)
    yv = np.linalg.cholesky(Kt) @ rng.standard_normal(100)
    g2 = ref.gaussian_process(yv, X, kernel='RBF2D', y_err=e)
    guess = np.array([1.0, 2.5, 2.5, 0.5])
    with quiet():
        g2.find_hyperparameters(hyperparameter_guess=guess.copy(), svd_method=False)
        fit_hyp = np.array(g2.hyperparameters, dtype=float)
        g2.compute_log_likelihood(fit_hyp, svd_method=False)
        ll_fit = float(g2.log_likelihood[0])
        g2.compute_log_likelihood(guess, svd_method=False)
        ll_guess = float(g2.log_likelihood[0])
    save('fit_2d_n100', x=X, y=yv, e=e, guess=guess, fit_hyp=fit_hyp, loglik_at_fit=ll_fit, loglik_at_guess=ll_guess)

    # ---------------- pull.py in 2D (N = 20) and for 3 ragged 1D objects ---
    X = rng.uniform(0., 10., (20, 2))
    e = np.full(20, 0.15)
    hyp2 = np.array([1.0, 2.5, 2.0, 0.8])
    with quiet():
        Kt = ref.rbf_kernel_2d(X, hyp2, y_err=e)
        yv = np.linalg.cholesky(Kt) @ rng.standard_normal(20)
        bp = ref.build_pull([yv], [X], hyp2, nugget=0.02, y_err=[e], kernel='RBF2D')
        bp.compute_pull(svd_method=False)
    save('pull_2d_n20', x=X, y=yv, e=e, hyp=hyp2, nugget=0.02, pull=np.array(bp.pull), residual=np.array(bp.residual),
         pull_average=bp.pull_average, pull_std=bp.pull_std)
    ns = (14, 23, 9)
    xs = [np.sort(rng.uniform(0., 6., k)) for k in ns]
    es = [rng.uniform(0.1, 0.25, k) for k in ns]
    ys = [smooth_1d(rng, xi, 0.15) for xi in xs]
    b3 = ref.build_pull(ys, xs, np.array([0.9, 1.2]), nugget=0.01, y_err=es)
    b3.compute_pull(svd_method=False)
    b3s = ref.build_pull(ys, xs, np.array([0.9, 1.2]), nugget=0.01, y_err=es)
    b3s.compute_pull(svd_method=False, diff=[0.1, -0.2, 0.05])
    pack = {}
    for i in range(3):
        pack['x_%d' % i], pack['y_%d' % i], pack['e_%d' % i] = xs[i], ys[i], es[i]
    save('pull_nobject_1d', nobj=3, hyp=np.array([0.9, 1.2]), nugget=0.01, pull=np.array(b3.pull),
         residual=np.array(b3.residual), pull_average=b3.pull_average, pull_std=b3.pull_std,
         diff=np.array([0.1, -0.2, 0.05]), pull_diff=np.array(b3s.pull), residual_diff=np.array(b3s.residual), **pack)

    # ---------------- a5 on a noise-free matrix: svd_method=True prediction, dropped modes ----------------
    # sigma = 0.0043 puts the largest singular value at 3.2e-4, so the reference's ABSOLUTE 1e-15 threshold sits at 3e-12 of it:
    # four orders above the rounding noise of LAPACK's decomposition (and of a correct Jacobi eigensolver).  Which modes pass
    # is then a property of the matrix (checked below: no singular value within a factor 3 of the threshold, and the
    # threshold above 1000 eps |K|)
    x = np.linspace(0., 5., 40)
    hyp = np.array([0.0043, 1.0])
    K = ref.rbf_kernel_1d(x, hyp)
    sv = np.linalg.svd(K, compute_uv=False)
    assert np.min(np.abs(np.log10(sv) + 15.)) > np.log10(3.), sv
    assert 1e-15 > 1000. * np.finfo(float).eps * sv[0]
    w = np.exp(-0.5 * (x - 2.2) ** 2) - 0.6 * np.exp(-0.5 * ((x - 3.6) / 0.7) ** 2)
    y = hyp[0] * w                                 # a smooth function of the kernel's own scale
    xn = np.linspace(0.1, 4.9, 17)
    gs = ref.gaussian_process(y, x, kernel='RBF1D')
    gs.hyperparameters = hyp
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gs.get_prediction(new_binning=xn, svd_method=True)
        inv, logdet = ref.svd_inverse(K, return_logdet=True)
    save('svd_noise_free_n40', x=x, y=y, xnew=xn, hyp=hyp, singular_values=sv, kept=int(np.sum(sv > 1e-15)),
         dropped=int(np.sum(sv <= 1e-15)), pred=gs.Prediction[0].copy(), cov_diag=np.diag(gs.covariance_matrix[0]).copy(),
         logdet=float(logdet), printed=np.array(buf.getvalue()))


if __name__ == '__main__':
    if '--round3' in sys.argv:
        main_round3()
    else:
        main()
