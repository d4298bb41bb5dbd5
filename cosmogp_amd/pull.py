"""Leave-one-out pulls -- mirror of cosmogp/pull.py (the main caller
of the hot path).

The reference re-builds a GP on N-1 points and predicts, N times per object, one after the
other (pull.py:51-90: O(N^4)).  Here ONE factorisation of each object's full matrix gives all its
leave-one-out predictions (cgp_batch_loo; the identities are in include/cosmogp_hip.h): O(N^3) per object,
objects bucketed by padded size and batched.  The literal form -- every (object, left-out point) pair as its
own sub-problem in batched launch sequences (cgp_batch_fit with one prediction point each) -- is kept
(`LOO_BY_REFITTING = True`, and as the fallback when the full matrix is not positive definite); the two agree
to rounding (tests/test_gpu_parity.py).  The per-subset mean offsets are the reference's (mean.py:95-97).
"""
import numpy as np
from scipy.stats import norm as normal

from . import _lib
from . import kernel as _kernel
import sys

from .gaussian_process import gaussian_process, _kparams

gpm = sys.modules['cosmogp_amd.gaussian_process']      # (the package re-exports a class of that name)
from .mean import return_mean

#: True: leave-one-out by refitting every sub-problem (the reference's own arithmetic, O(N^4)); False: from one
#: factorisation of the full matrix per object
LOO_BY_REFITTING = False


class build_pull(object):
    """Pull / residual builder with the reference's constructor and attributes (pull.py:8-40)."""

    def __init__(self, y, x, hyperparameters, nugget=0., y_err=None, y_mean=None,
                 x_axis_mean=None, kernel='RBF1D', device=None):
        self.y = y
        self.x = x
        self.hyperparameters = hyperparameters
        self.y_err = y_err
        self.y_mean = y_mean
        self.x_axis_mean = x_axis_mean
        self.kernel = kernel
        self.nugget = nugget
        self.n_object = len(y)
        self._pull = []
        self.pull = []
        self.residual = []
        self.prediction = []
        self.pull_average = None
        self.pull_std = None
        self._device = device

    def compute_pull(self, diff=None, svd_method=True, substract_mean=False):
        """Fill self.pull / self.residual / self.prediction (pull.py:43-102).

        The leave-one-out sub-problems of ALL objects (sum of n_i of them) are bucketed by padded size and run as
        batched launch sequences with one prediction point each.  svd_method=True (the reference's default) takes
        the same batched Cholesky route whenever the noise diagonal proves that the pseudo-inverse is the inverse
        (gaussian_process.SVD_VIA_CHOLESKY_WHEN_PD, 'auto' by default), otherwise -- or when a sub-matrix turns
        out not to be positive definite -- the sub-problems go one by one through the device eigensolver route."""
        dev = self._device if self._device is not None else _lib.default_device()
        dim = 1 if self.kernel == 'RBF1D' else 2
        objs = []
        for sn in range(self.n_object):
            y = np.asarray(self.y[sn], dtype=np.float64)
            x = np.asarray(self.x[sn], dtype=np.float64)
            yerr = np.zeros_like(y) if self.y_err is None else np.asarray(self.y_err[sn], dtype=np.float64)
            dd = None if diff is None else diff[sn]
            if self.y_mean is None and substract_mean:                       # pull.py:71-73
                self.y_mean = np.ones_like(y) * np.mean(y)
                self.x_axis_mean = x
            objs.append((y, x, yerr, dd))
        batched = not svd_method
        if svd_method and gpm.SVD_VIA_CHOLESKY_WHEN_PD:
            bound = min(float(np.min(o[2] ** 2)) if len(o[0]) else np.inf for o in objs) + float(self.nugget) ** 2
            batched = gpm.SVD_VIA_CHOLESKY_WHEN_PD is True or bound > 1e-14
        results = None
        if batched and not LOO_BY_REFITTING:
            try:
                results = self._loo_from_full_factor(dev, dim, objs, substract_mean)
            except np.linalg.LinAlgError:
                results = None                 # e.g. two coincident points without noise: the sub-problems may still be fine
        if batched and results is None:
            try:
                results = self._loo_batched(dev, dim, objs, substract_mean)
            except np.linalg.LinAlgError:
                if not svd_method:
                    raise
        if results is None:
            results = [self._loo_svd(sn, o[0], o[1], o[2], o[3], substract_mean) for sn, o in enumerate(objs)]
        for sn, (y, x, yerr, dd) in enumerate(objs):
            pred, pred_var = results[sn]
            pull = (pred - y)
            pull /= np.sqrt(yerr ** 2 + pred_var + self.nugget ** 2)
            res = pred - y
            self._pull.append(pull)
            self.prediction.append(pred)
            self.pull.extend(pull)                 # (element by element like pull.py:96-98)
            self.residual.extend(res)
        self.pull_average, self.pull_std = normal.fit(self.pull)

    def _loo_from_full_factor(self, dev, dim, objs, substract_mean):
        """All leave-one-out predictions of an object from the factor of its FULL matrix (cgp_batch_loo):
        with r0 = y - mean shape, alpha = K^-1 r0, beta = K^-1 1, g = diag K^-1 and the left-out point t,
            k_t' K_-t^-1 (r0 - diff_t)_-t = (r0_t - alpha_t / g_t) - diff_t (1 - beta_t / g_t)
            k_t' K_-t^-1 k_t             = K_tt - 1 / g_t
        where diff_t = mean of (y - shape) over the kept points (mean.py:95-97) unless `diff` is given.  The cross row of
        the reference is amp_cross / amp_train times the train row (kernel.py:137-151: the 2D train matrix carries no
        amplitude), and K(new, new)_tt = amp_train + nugget^2 as in get_prediction.  Returns [(pred, pred_var)] per object."""
        use_mean = substract_mean or self.y_mean is not None
        kp = _kparams(dim, self.hyperparameters, self.nugget)
        rho = kp[1] / kp[0]
        kss = kp[0] + float(self.nugget) ** 2
        out = [None] * len(objs)
        buckets = {}
        for sn, (y, x, yerr, dd) in enumerate(objs):
            n = len(y)
            if n < 2:
                raise np.linalg.LinAlgError('leave-one-out needs at least two points per object')
            if use_mean:
                # shape(x) of the FIXED mean function; the offset of a subset is shape + diff (return_mean with diff = 0)
                shape = np.atleast_1d(return_mean(y, x, mean_y=self.y_mean, mean_xaxis=self.x_axis_mean, diff=0.) * np.ones(n))
            else:
                shape = np.zeros(n)
            buckets.setdefault((n + 127) // 128 * 128, []).append((sn, shape))
        for np_ in sorted(buckets, reverse=True):
            members = buckets[np_]
            per = max(1, min(gpm.MAX_BATCH_OBJECTS, int(gpm.BATCH_BYTES_LIMIT // ((3 * np_ + 128) * np_ * 8))))
            for lo in range(0, len(members), per):
                part = members[lo:lo + per]
                r0s = [objs[sn][0] - shape for sn, shape in part]
                dev.batch_set_data(dim, [objs[sn][1] for sn, _ in part], r0s,
                                   [_kernel.diag_term(len(objs[sn][0]), objs[sn][2], 0., 0.) for sn, _ in part])
                alpha, beta, g, info = dev.batch_loo(kp)
                if np.any(info):
                    b = int(np.nonzero(info)[0][0])
                    raise np.linalg.LinAlgError('matrix of object %d is not positive definite' % part[b][0])
                # the whole chunk at once on (objects, padded size) arrays; columns beyond an object's n are scratch
                nb_, npad = len(part), alpha.shape[1]
                nn = np.array([len(objs[sn][0]) for sn, _ in part])
                valid = np.arange(npad)[None, :] < nn[:, None]
                R0, E2, SH = np.zeros((nb_, npad)), np.zeros((nb_, npad)), np.zeros((nb_, npad))
                for k, (sn, shape) in enumerate(part):
                    R0[k, :nn[k]] = r0s[k]
                    E2[k, :nn[k]] = objs[sn][2] ** 2
                    SH[k, :nn[k]] = shape
                gg = np.where(valid, g, 1.)
                if not use_mean:
                    DT = np.zeros((nb_, npad))
                else:
                    DT = (R0.sum(axis=1)[:, None] - R0) / (nn[:, None] - 1.)          # mean over the kept points
                    for k, (sn, _) in enumerate(part):
                        dd = objs[sn][3]
                        if dd is not None:
                            DT[k, :] = float(np.ravel(dd)[0]) if np.ndim(dd) else float(dd)
                MU = rho * ((R0 - alpha / gg) - DT * (1. - beta / gg))
                VAR = np.abs(kss - rho * rho * ((kp[0] + (E2 + kp[6])) - 1. / gg))
                if use_mean:
                    MU = MU + (SH + DT)
                for k, (sn, _) in enumerate(part):
                    out[sn] = (MU[k, :nn[k]].copy(), VAR[k, :nn[k]].copy())
        return out

    def _loo_batched(self, dev, dim, objs, substract_mean):
        """Every (object, left-out point) pair is one sub-problem; returns [(pred, pred_var)] per object."""
        use_mean = substract_mean or self.y_mean is not None
        subs = {}                              # padded size -> list of (sn, t, x_keep, r_keep, d_keep)
        offs = [np.zeros(len(o[0])) for o in objs]
        for sn, (y, x, yerr, dd) in enumerate(objs):
            n = len(y)
            for t in range(n):
                keep = np.ones(n, dtype=bool)
                keep[t] = False
                if use_mean:
                    y0 = return_mean(y[keep], x[keep], mean_y=self.y_mean, mean_xaxis=self.x_axis_mean, diff=dd)
                    offs[sn][t] = np.atleast_1d(return_mean(y[keep], x[keep], new_x=x, mean_y=self.y_mean,
                                                            mean_xaxis=self.x_axis_mean, diff=dd) * np.ones(n))[t]
                else:
                    y0 = 0.
                np_ = (max(n - 1, 1) + 127) // 128 * 128
                subs.setdefault(np_, []).append((sn, t, x[keep], y[keep] - y0,
                                                 _kernel.diag_term(n - 1, yerr[keep], 0., 0.)))
        out = [(np.zeros(len(o[0])), np.zeros(len(o[0]))) for o in objs]
        kp = _kparams(dim, self.hyperparameters, self.nugget)
        kss_noise = 0. ** 2 + 0. ** 2 + self.nugget ** 2
        for np_ in sorted(subs, reverse=True):
            members = subs[np_]
            per = max(1, min(gpm.MAX_BATCH_OBJECTS, int(gpm.BATCH_BYTES_LIMIT // ((np_ + 128) * np_ * 8))))
            for lo in range(0, len(members), per):
                part = members[lo:lo + per]
                if any(len(p[3]) == 0 for p in part):
                    raise np.linalg.LinAlgError('leave-one-out needs at least two points per object')
                grids = np.stack([np.asarray(objs[p[0]][1][p[1]], dtype=np.float64).reshape((1,) + objs[p[0]][1].shape[1:])
                                  for p in part])
                _, info, mu, var = dev.batch_fit(dim, [p[2] for p in part], [p[3] for p in part], [p[4] for p in part],
                                                 kp, xnew=grids, kss_noise=kss_noise, xnew_per_object=True)
                if np.any(info):
                    b = int(np.nonzero(info)[0][0])
                    raise np.linalg.LinAlgError('leave-one-out matrix (object %d, point %d) is not positive definite'
                                                % (part[b][0], part[b][1]))
                for k, p in enumerate(part):
                    out[p[0]][0][p[1]] = mu[k, 0] + offs[p[0]][p[1]]
                    out[p[0]][1][p[1]] = abs(var[k, 0])
        return out

    def _loo_svd(self, sn, y, x, yerr, dd, substract_mean):
        """The reference's own loop (pull.py:51-90) through the GP class with svd_method=True forced onto the
        device eigensolver route (used for noise-free / non-positive-definite data)."""
        n = len(y)
        pred, pred_var = np.zeros(n), np.zeros(n)
        saved = gpm.SVD_VIA_CHOLESKY_WHEN_PD
        try:
            gpm.SVD_VIA_CHOLESKY_WHEN_PD = False
            for t in range(n):
                keep = np.ones(n, dtype=bool)
                keep[t] = False
                gpp = gaussian_process(y[keep], x[keep], y_err=yerr[keep], Mean_Y=self.y_mean,
                                       Time_mean=self.x_axis_mean, kernel=self.kernel,
                                       diff=None if dd is None else [dd], substract_mean=substract_mean,
                                       device=self._device)
                gpp.hyperparameters = self.hyperparameters
                gpp.nugget = self.nugget
                gpp.get_prediction(new_binning=x, svd_method=True)
                self.gpp = gpp
                pred[t] = gpp.Prediction[0][t]
                pred_var[t] = abs(gpp.covariance_matrix[0][t, t])
        finally:
            gpm.SVD_VIA_CHOLESKY_WHEN_PD = saved
        return pred, pred_var

    def plot_result(self, binning=60):
        """Histogram of the pulls with the fitted normal law (pull.py:105-140)."""
        import pylab as plt
        plt.figure()
        plt.hist(self.pull, bins=binning, density=True)
        xmin, xmax = plt.xlim()
        lim = max([abs(xmin), abs(xmax)])
        plt.xlim(-lim, lim)
        grid = np.linspace(-lim, lim, 100)
        plt.plot(grid, normal.pdf(grid, self.pull_average, self.pull_std), 'r', linewidth=3)
        npull = len(self.pull)
        plt.title(r"Fit results: $\mu$ = $ %.2f \pm %.2f $, $\sigma$ = $ %.2f \pm %.2f $"
                  % (self.pull_average, self.pull_std / np.sqrt(npull),
                     self.pull_std, self.pull_std / np.sqrt(2 * npull)))
        plt.ylabel('Number of points (normed)')
        plt.xlabel('Pull')
