"""Leave-one-out pulls -- mirror of cosmogp/pull.py (the main caller
of the hot path).

The reference re-builds a GP on N-1 points and predicts, N times per object, one after the
other (pull.py:51-90: O(N^4)).  Here the N leave-one-out problems of an object are ONE
batched launch sequence on the device (cgp_batch_fit with one prediction point per
sub-problem), the per-subset mean offsets are the same host O(N) code as the reference.
svd_method=True keeps the reference's host SVD route through the Gaussian_process class.
"""
import numpy as np
from scipy.stats import norm as normal

from . import _lib
from . import kernel as _kernel
import sys

from .gaussian_process import gaussian_process, _kparams

gpm = sys.modules['cosmogp_amd.gaussian_process']      # (the package re-exports a class of that name)
from .mean import return_mean


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
        """Fill self.pull / self.residual / self.prediction (pull.py:43-102)."""
        dev = self._device if self._device is not None else _lib.default_device()
        dim = 1 if self.kernel == 'RBF1D' else 2
        for sn in range(self.n_object):
            y = np.asarray(self.y[sn], dtype=np.float64)
            x = np.asarray(self.x[sn], dtype=np.float64)
            n = len(y)
            yerr = np.zeros_like(y) if self.y_err is None else np.asarray(self.y_err[sn], dtype=np.float64)
            dd = None if diff is None else diff[sn]
            if self.y_mean is None and substract_mean:                       # pull.py:71-73
                self.y_mean = np.ones_like(y) * np.mean(y)
                self.x_axis_mean = x
            use_mean = substract_mean or self.y_mean is not None
            if svd_method:
                pred, pred_var = self._loo_host_svd(sn, y, x, yerr, dd, substract_mean)
            else:
                xs, rs, ds, offs = [], [], [], np.zeros(n)
                for t in range(n):
                    keep = np.ones(n, dtype=bool)
                    keep[t] = False
                    if use_mean:
                        y0 = return_mean(y[keep], x[keep], mean_y=self.y_mean, mean_xaxis=self.x_axis_mean,
                                         diff=dd)
                        offs[t] = np.atleast_1d(return_mean(y[keep], x[keep], new_x=x, mean_y=self.y_mean,
                                                            mean_xaxis=self.x_axis_mean, diff=dd)
                                                * np.ones(n))[t]
                    else:
                        y0 = 0.
                    xs.append(x[keep])
                    rs.append(y[keep] - y0)
                    ds.append(_kernel.diag_term(n - 1, yerr[keep], 0., 0.))
                grids = x.reshape((n, 1) + x.shape[1:])                      # one point per sub-problem
                # the n sub-problems go down in chunks whose padded batch fits the HBM budget
                np_ = (n - 1 + 127) // 128 * 128
                per = max(1, int(gpm.BATCH_BYTES_LIMIT // ((np_ + 128) * np_ * 8)))
                pred, pred_var = np.zeros(n), np.zeros(n)
                kp = _kparams(dim, self.hyperparameters, self.nugget)
                for lo in range(0, n, per):
                    hi = min(n, lo + per)
                    _, info, mu, var = dev.batch_fit(dim, xs[lo:hi], rs[lo:hi], ds[lo:hi], kp, xnew=grids[lo:hi],
                                                     kss_noise=0. ** 2 + 0. ** 2 + self.nugget ** 2,
                                                     xnew_per_object=True)
                    if np.any(info):
                        raise np.linalg.LinAlgError('leave-one-out matrix %d is not positive definite'
                                                    % (lo + int(np.nonzero(info)[0][0])))
                    pred[lo:hi] = mu[:, 0] + offs[lo:hi]
                    pred_var[lo:hi] = np.abs(var[:, 0])
            pull = (pred - y)
            pull /= np.sqrt(yerr ** 2 + pred_var + self.nugget ** 2)
            res = pred - y
            self._pull.append(pull)
            self.prediction.append(pred)
            for t in range(n):
                self.pull.append(pull[t])
                self.residual.append(res[t])
        self.pull_average, self.pull_std = normal.fit(self.pull)

    def _loo_host_svd(self, sn, y, x, yerr, dd, substract_mean):
        n = len(y)
        pred, pred_var = np.zeros(n), np.zeros(n)
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
