"""TEST INFRASTRUCTURE: a stand-in for cosmogp_amd._lib.Device built on the CPU oracle, used only
by the CPU tests of the host logic (sharding over ranks, batching, lazy attributes) where no GPU
exists.  It is never importable from the product package."""
import numpy as np

from oracle import cosmogp_oracle as orc


def _kernel_from_kp(dim, kp):
    """Rebuild oracle-callable kernels from a kp vector (include/cosmogp_hip.h)."""
    kp = np.asarray(kp, dtype=np.float64)

    def cross_or_train(x, xnew=None, diag=None):
        x = np.asarray(x, dtype=np.float64)
        other = x if xnew is None else np.asarray(xnew, dtype=np.float64)
        if dim == 1:
            d = x[None, :] - other[:, None]
            arg = -0.5 * ((d * d) / kp[2])
        else:
            vi = kp[2:6].reshape(2, 2)
            dist = orc._mahalanobis_sq_then_sqrt(x, other, vi).T
            arg = -0.5 * dist ** 2
        if xnew is None:
            k = kp[0] * np.exp(arg)
            idx = np.arange(len(x))
            k[idx, idx] = kp[0] + ((0. if diag is None else diag) + kp[6])
            return k
        return kp[1] * np.exp(arg)

    return cross_or_train


class FakeDevice(object):
    def __init__(self):
        self.generation = 0
        self.batch_generation = 0
        self.calls = []
        self.diag = None

    def rbf(self, dim, x, kp, xnew=None, diag=None):
        return _kernel_from_kp(dim, kp)(x, xnew, diag)

    def set_data(self, dim, x, resid):
        self.generation += 1
        self.dim, self.x, self.r, self.n = dim, np.asarray(x, float), np.asarray(resid, float), len(resid)
        self.diag = None
        self.calls.append('set_data')

    def _one(self, dim, x, r, diag, kp, xnew, mean_new, kss_noise):
        kern = _kernel_from_kp(dim, kp)
        K = kern(x, None, diag)
        try:
            if xnew is None:
                _, _, ll = orc.fit_predict_cho(K, r, np.zeros((0, len(r))), np.zeros(0))
                return ll, None, None
            H = kern(x, xnew)
            mu, var, ll = orc.fit_predict_cho(K, r, H, np.full(len(H), kp[0] + kss_noise))
        except np.linalg.LinAlgError:
            raise
        return ll, mu + (0. if mean_new is None else mean_new), var

    def fit(self, kp, diag=None, xnew=None, mean_new=None, kss_noise=0., want_var=True):
        self.calls.append('fit')
        if diag is not None:
            self.diag = np.array(diag)          # resident, like the real handle
        self.kp = np.array(kp)
        ll, mu, var = self._one(self.dim, self.x, self.r, self.diag, kp, xnew, mean_new, kss_noise)
        return ll, 0., mu, var

    def svd_inverse(self, K, thresh=1e-15):
        inv, logdet = orc.svd_inverse(np.asarray(K, float), return_logdet=True)
        return inv, logdet, 0

    def fit_svd(self, kp, diag=None, xnew=None, mean_new=None, kss_noise=0., thresh=1e-15, want_cov=False):
        self.calls.append('fit_svd')
        if diag is not None:
            self.diag = np.array(diag)
        self.kp = np.array(kp)
        kern = _kernel_from_kp(self.dim, kp)
        K = kern(self.x, None, self.diag)
        inv, logdet = orc.svd_inverse(K, return_logdet=True)
        n = len(self.r)
        ll = -0.5 * self.r @ inv @ self.r - 0.5 * n * np.log(2 * np.pi) - 0.5 * logdet
        if xnew is None:
            return ll, logdet, 0, None, None, None
        H = kern(self.x, xnew)
        mu = H @ inv @ self.r + (0. if mean_new is None else mean_new)
        kp0 = self.kp.copy()
        kp0[6] = 0.
        cov = _kernel_from_kp(self.dim, kp0)(np.asarray(xnew, float), None, np.full(len(H), kss_noise)) - H @ inv @ H.T
        return ll, logdet, 0, mu, np.diag(cov).copy(), cov if want_cov else None

    def batch_fit_grad_resident(self, kp):
        dim, xs, resids, diags = self._b
        self.calls.append('batch_fit_grad:%d' % len(xs))
        B = len(xs)
        ll, info, sums = np.zeros(B), np.zeros(B, dtype=np.int32), np.zeros((B, 5))
        keep = (getattr(self, 'dim', None), getattr(self, 'x', None), getattr(self, 'r', None), self.diag)
        for b in range(B):
            self.dim, self.x, self.r, self.n = dim, xs[b], resids[b], len(resids[b])
            self.diag = None if diags is None else diags[b]
            try:
                ll[b], _, sums[b] = self.fit_grad(kp)
            except np.linalg.LinAlgError:
                info[b] = 1
        self.dim, self.x, self.r, self.diag = keep
        return ll, info, sums

    def sweep_loglik(self, kps, diag=None):
        self.calls.append('sweep_loglik:%d' % len(kps))
        self.batch_generation += 1
        if diag is not None:
            self.diag = np.array(diag)
        ll = np.zeros(len(kps))
        info = np.zeros(len(kps), dtype=np.int32)
        for c, kp in enumerate(kps):
            try:
                ll[c] = self._one(self.dim, self.x, self.r, self.diag, kp, None, None, 0.)[0]
            except np.linalg.LinAlgError:
                info[c] = 1
        return ll, info

    def fit_grad(self, kp, diag=None):
        """Moment sums of include/cosmogp_hip.h (cgp_fit_grad), from dense NumPy."""
        self.calls.append('fit_grad')
        if diag is not None:
            self.diag = np.array(diag)
        self.kp = np.array(kp)
        kern = _kernel_from_kp(self.dim, kp)
        K = kern(self.x, None, self.diag)
        _, _, ll = orc.fit_predict_cho(K, self.r, np.zeros((0, len(self.r))), np.zeros(0))
        Ki = np.linalg.inv(K)
        a = Ki @ self.r
        M = np.outer(a, a) - Ki
        E = (K - np.diag(np.diag(K)) + kp[0] * np.eye(len(K))) / kp[0]
        X = self.x.reshape(len(K), -1)
        d0 = X[None, :, 0] - X[:, None, 0]
        d1 = (X[None, :, 1] - X[:, None, 1]) if self.dim == 2 else np.zeros_like(d0)
        sums = np.array([np.sum(M * E), np.sum(M * E * d0 * d0), np.sum(M * E * d0 * d1), np.sum(M * E * d1 * d1),
                         np.trace(M)])
        return ll, 0., sums

    def predict(self, xnew, mean_new=None, kss_noise=0., want_cov=False):
        kern = _kernel_from_kp(self.dim, self.kp)
        K = kern(self.x, None, self.diag)
        H = kern(self.x, xnew)
        Ki = np.linalg.inv(K)
        kp0 = self.kp.copy()
        kp0[6] = 0.
        cov = _kernel_from_kp(self.dim, kp0)(np.asarray(xnew, float), None, np.full(len(H), kss_noise)) - H @ Ki @ H.T
        mu = H @ Ki @ self.r + (0. if mean_new is None else mean_new)
        return mu, np.diag(cov).copy(), cov if want_cov else None

    def fetch(self, which):
        K = _kernel_from_kp(self.dim, self.kp)(self.x, None, self.diag)
        return [K, np.linalg.cholesky(K), np.linalg.inv(K)][which]

    def batch_set_data(self, dim, xs, resids, diags=None):
        self.batch_generation += 1
        self.calls.append('batch_set_data:%d' % len(xs))
        self._b = (dim, [np.asarray(a, float) for a in xs], [np.asarray(a, float) for a in resids],
                   None if diags is None else [np.asarray(a, float) for a in diags])

    def batch_fit_resident(self, kp, xnew=None, mean_new=None, kss_noise=0., xnew_per_object=False):
        dim, xs, resids, diags = self._b
        return self._batch(dim, xs, resids, diags, kp, xnew, mean_new, kss_noise, xnew_per_object)

    def batch_fit(self, dim, xs, resids, diags, kp, xnew=None, mean_new=None, kss_noise=0.,
                  xnew_per_object=False):
        self.batch_set_data(dim, xs, resids, diags)
        return self.batch_fit_resident(kp, xnew, mean_new, kss_noise, xnew_per_object)

    def _batch(self, dim, xs, resids, diags, kp, xnew, mean_new, kss_noise, xnew_per_object):
        self.calls.append('batch_fit:%d' % len(xs))
        B = len(xs)
        ll, info = np.zeros(B), np.zeros(B, dtype=np.int32)
        mu = var = None
        for b in range(B):
            grid = None if xnew is None else (np.asarray(xnew)[b] if xnew_per_object else xnew)
            try:
                l, m_, v_ = self._one(dim, np.asarray(xs[b], float), np.asarray(resids[b], float),
                                      None if diags is None else diags[b], kp, grid, mean_new, kss_noise)
            except np.linalg.LinAlgError:
                info[b] = 1
                continue
            ll[b] = l
            if m_ is not None:
                if mu is None:
                    mu, var = np.zeros((B, len(m_))), np.zeros((B, len(m_)))
                mu[b], var[b] = m_, v_
        return ll, info, mu, var


def _fake_bispev(self, tck, points):
    """oracle-side evaluation of a bivariate spline (scipy's bisplev per point, like cosmogp/mean.py:63-65)"""
    import scipy.interpolate as inter
    self.calls.append('bispev:%d' % len(points))
    pts = np.asarray(points, dtype=np.float64)
    return np.array([inter.bisplev(p[0], p[1], tck) for p in pts])


FakeDevice.bispev = _fake_bispev
