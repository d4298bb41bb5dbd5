"""GP regressor objects -- host mirror of cosmogp/Gaussian_process.py.

Same class names, constructor arguments, methods and result attributes as the reference
(Gaussian_process.py:78-393), so scripts written for cosmogp run unchanged.  What differs
is where the O(N^2)/O(N^3) work happens:

  svd_method=False  the accelerated path.  Per object ONE call into libcosmogp_hip
                    (cgp_fit): K is assembled in HBM, factored by the blocked MFMA
                    Cholesky, and the cross matrix + residual ride along as extra rows,
                    so log-likelihood, prediction and predictive variance come back as
                    O(M) vectors.  N x N matrices never cross PCIe unless a caller reads
                    .kernel_matrix / .inv_kernel_matrix / .covariance_matrix, which are
                    lazy lists filled from the device on first access.
  svd_method=True   the reference's default algorithm (pseudo-inverse from the SVD of K with
                    modes <= 1e-15 dropped, inv_matrix.py:4-18), kept for its semantics on
                    singular K.  Also ONE device call per object (cgp_fit_svd): K is symmetric, so
                    its SVD is its eigen-decomposition, computed in HBM by a block Jacobi
                    eigensolver (csrc/jacobi.hpp); K^+, H K^+ r and K** - H K^+ H' follow with
                    the MFMA GEMM.  O(20 N^3): use svd_method=False when K is positive definite.

Documented deviations from the reference (SURVEY.md 8a "quirks"):
  * get_prediction with new_binning=None uses each object's own grid; the reference
    indexes with a stale loop variable (Gaussian_process.py:330) and so uses the LAST
    object's grid for every object.  REFERENCE_STALE_INDEX=True restores that.
  * fit_nugget exists (False) from construction; the reference only creates it inside
    find_hyperparameters (:242-245), so compute_log_likelihood first raises AttributeError.
  * the 2D train-train sigma^2 omission is kept by default (kernel.SIGMA2_IN_TRAIN_2D).
"""
import numpy as np
from scipy.optimize import fmin

from . import _lib
from . import kernel as _kernel
from .inv_matrix import svd_inverse, SVD_THRESHOLD
from .mean import return_mean

REFERENCE_STALE_INDEX = False

#: svd_method=True (the reference's default) means: pseudo-inverse from the decomposition of K with the
#: modes whose singular value is <= 1e-15 dropped (inv_matrix.py:4-18).  How it is computed:
#:   False   always by the device eigensolver (cgp_fit_svd), the reference's algorithm literally;
#:   'auto'  (default) K = K_rbf + diag(y_err^2 + nugget^2) with K_rbf positive semi-definite, so every
#:           eigenvalue of K is >= min(y_err^2) + nugget^2: when that bound exceeds 1e-14 NO mode can be
#:           dropped, the pseudo-inverse IS the inverse, and the call is served by the device Cholesky
#:           route (same numbers to ~cond(K) eps, ~1000x faster); otherwise, or when the factorisation
#:           meets a non-positive pivot, by the eigensolver;
#:   True    try the Cholesky route first whatever the bound says, eigensolver on a non-positive pivot.
SVD_VIA_CHOLESKY_WHEN_PD = 'auto'


#: objects per batched launch sequence are capped so that B * (Np + 128) * Np * 8 bytes
#: stays below this (HBM is 288 GB; leave room for several handles)
BATCH_BYTES_LIMIT = 64 << 30
#: the batch index travels in gridDim.z (<= 65535)
MAX_BATCH_OBJECTS = 65535


class _LazyList(object):
    """List-like whose items are produced (and cached) on first access: keeps the
    reference's `gp.covariance_matrix[0]` / `gp.inv_kernel_matrix[sn]` idiom without moving
    N x N matrices over PCIe when nobody looks at them."""

    def __init__(self, n, producer):
        self._n = n
        self._producer = producer
        self._items = {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._items:
            self._items[i] = self._producer(i)
        return self._items[i]

    def __setitem__(self, i, value):
        self._items[i] = value

    def __iter__(self):
        return (self[k] for k in range(self._n))


def log_likelihood_gp(y, x_axis, kernel, hyperparameter, nugget, y_err=None, y_mean=None,
                      svd_method=True, device=None):
    """Log marginal likelihood of one object; same contract as Gaussian_process.py:13-75
    (returns an ndarray of shape (1,)).  `kernel` is 'RBF1D' / 'RBF2D' or one of this
    package's rbf_kernel_1d / rbf_kernel_2d callables."""
    dim = _kernel_dim(kernel)
    if y_mean is None:
        y_mean = np.zeros_like(y)
    y = np.asarray(y, dtype=np.float64)
    resid = y - y_mean
    n = len(x_axis)
    dev = device if device is not None else _lib.default_device()
    if svd_method:
        # reference default: pseudo-inverse route (Gaussian_process.py:54-57), all on the device
        dev.set_data(dim, x_axis, np.ones(n) * resid)
        ll = dev.fit_svd(_kparams(dim, hyperparameter, nugget), _kernel.diag_term(n, y_err, 0., 0.),
                         thresh=SVD_THRESHOLD)[0]
        return np.array([ll])
    dev.set_data(dim, x_axis, np.ones(n) * resid)
    ll, _, _, _ = dev.fit(_kparams(dim, hyperparameter, nugget), _kernel.diag_term(n, y_err, 0., 0.))
    return np.array([ll])


def _kernel_dim(kernel):
    if kernel in ('RBF1D', _kernel.rbf_kernel_1d):
        return 1
    if kernel in ('RBF2D', _kernel.rbf_kernel_2d):
        return 2
    raise ValueError('kernel must be RBF1D / RBF2D (or rbf_kernel_1d / rbf_kernel_2d)')


def _kparams(dim, hyperparameter, nugget=0.):
    if dim == 1:
        return _kernel.kernel_params_1d(hyperparameter, nugget)
    return _kernel.kernel_params_2d(hyperparameter, nugget=nugget)


class Gaussian_process(object):
    """Gaussian process regressor over a list of independent objects
    (mirror of cosmogp.Gaussian_process, Gaussian_process.py:78-361)."""

    def __init__(self, y, Time, kernel='RBF1D', y_err=None, diff=None, Mean_Y=None,
                 Time_mean=None, substract_mean=False, device=None):
        kernel_choice = ['RBF1D', 'RBF2D']
        assert kernel in kernel_choice, '%s is not in implemented kernel' % (kernel)
        self._dim = 1 if kernel == 'RBF1D' else 2
        self._device = device
        self.comm = None            # set to multigpu.Comm() to shard objects over ranks
        if self._dim == 1:
            self.kernel = _kernel.rbf_kernel_1d
            sigma, L = _kernel.init_rbf(Time, y)
            self.hyperparameters = np.array([sigma, L])
        else:
            self.kernel = _kernel.rbf_kernel_2d
            sigma, L = _kernel.init_rbf(Time, y)
            self.hyperparameters = np.array([sigma, L, L, 0.])
        self.y = y
        self.N_sn = len(y)
        self.Time = Time
        self.nugget = 0.
        self.fit_nugget = False
        if y_err is not None:
            self.y_err = y_err
        else:
            self.y_err = [np.zeros(len(self.y[i])) for i in range(self.N_sn)]
        self.Mean_Y = Mean_Y
        self.Time_mean = Time_mean
        self.substract_mean = substract_mean
        self.diff = np.array([None] * self.N_sn) if diff is None else diff
        if self.substract_mean or self.Mean_Y is not None:
            self.y0 = [return_mean(self.y[i], self.Time[i], mean_y=self.Mean_Y,
                                   mean_xaxis=self.Time_mean, diff=self.diff[i])
                       for i in range(self.N_sn)]
        else:
            self.y0 = np.zeros(self.N_sn)
        self.as_the_same_time = True
        self._resident = None       # _res_token of the object whose data is in HBM
        self._factor_token = None   # + (kp, nugget) of the factor in HBM
        self._batch_resident = None
        self._cov_state = None

    # ------------------------------------------------------------------ helpers
    @property
    def device(self):
        if self._device is None:
            self._device = _lib.default_device()
        return self._device

    def _resid(self, sn):
        return np.asarray(self.y[sn], dtype=np.float64) - self.y0[sn]

    def _diag(self, sn):
        """y_err**2 + floor**2 (floor = 0); the nugget**2 term travels in kp[6]."""
        return _kernel.diag_term(len(self.y[sn]), self.y_err[sn], 0., 0.)

    def _svd_by_cholesky(self, nugget):
        """May a svd_method=True call be served by the Cholesky route? (see SVD_VIA_CHOLESKY_WHEN_PD)"""
        if SVD_VIA_CHOLESKY_WHEN_PD == 'auto':
            # min over every point of y_err**2: a property of the data, cached per identity of the y_err arrays (this is
            # asked once per objective evaluation: recomputing it cost ten times the device call for 1000 light curves)
            key = tuple(id(e) for e in self.y_err) if self.y_err is not None else None
            if getattr(self, '_noise_bound_key', ()) != key:
                self._noise_bound = min((float(np.min(self._diag(sn))) if len(self.y[sn]) else np.inf
                                         for sn in range(self.N_sn)), default=np.inf)
                self._noise_bound_key = key
            return self._noise_bound + float(nugget) ** 2 > 1e-14
        return bool(SVD_VIA_CHOLESKY_WHEN_PD)

    def _res_token(self, sn):
        """Identity of object sn's data in HBM: the handle's data generation, the index, the ids of the arrays it was uploaded
        from AND a fingerprint of their contents (length, sum, first / middle / last element of y, Time, y_err and y0 -- `diff` enters
        through y0 = mean + diff, it is not fingerprinted by itself):
        replacing gp.y[sn] / gp.Time[sn] / gp.y_err[sn] by new arrays is noticed, and so is the usual in-place edit
        (`gp.y_err[0] *= 2`, `gp.y[0][:] = ...`) -- the reference re-reads the arrays on every call
        (Gaussian_process.py:207-211).  An edit that leaves all of these unchanged needs invalidate().  (One O(n) sum per array
        and call: microseconds beside a fit.  The many-object batch token stays on identities: see _batch_upload.)"""
        def fp(a):
            a = np.asarray(a)
            if a.size == 0:
                return (0,)
            f = a.reshape(-1)
            return (f.size, float(f.sum()), float(f[0]), float(f[f.size // 2]), float(f[-1]))
        y0 = self.y0[sn] if self.y0 is not None else 0.
        return (self.device.generation, sn, id(self.y[sn]), id(self.Time[sn]), id(self.y_err[sn]),
                fp(self.y[sn]), fp(self.Time[sn]), fp(self.y_err[sn]), fp(y0))

    def _make_resident(self, sn):
        """Upload object sn unless it already is the resident object of the handle; returns the diagonal
        vector to pass along with the next fit (None = keep the resident one)."""
        dev = self.device
        if self._resident == self._res_token(sn):
            return None
        dev.set_data(self._dim, self.Time[sn], self._resid(sn))
        self._resident = self._res_token(sn)
        return self._diag(sn)

    def invalidate(self):
        """Forget what is resident in HBM (call after mutating y / y_err / Time / y0 IN PLACE: the reference
        re-reads them on every call, Gaussian_process.py:207-211; here they are uploaded once)."""
        self._resident = None
        self._factor_token = None
        self._batch_resident = None
        self._noise_bound_key = ()

    def _my_objects(self):
        if self.comm is None:
            return list(range(self.N_sn))
        return self.comm.my_units(self.N_sn)

    def _batch_chunks(self, objs, grad=False, m=0):
        """Split object indices into launch sequences: objects are bucketed by their padded size (multiples of
        128), so that one large object does not inflate the matrices of the small ones, and a bucket is cut where
        its batch would exceed BATCH_BYTES_LIMIT -- Np + round_up(m + 1, 128) rows of Np doubles per object for a
        fit with m prediction points, 3 Np + 128 for the gradient (K, the identity rows, K^-1) -- or
        MAX_BATCH_OBJECTS (the batch index is a grid dimension).  Chunks come largest size first; inside a
        chunk the object order is kept."""
        # (asked once per objective evaluation: the answer only depends on the lengths of the objects and the limits)
        key = (tuple(objs), tuple(map(len, self.y)), bool(grad), int(m), BATCH_BYTES_LIMIT, MAX_BATCH_OBJECTS)
        cached = getattr(self, '_chunks_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        buckets = {}

        def small_class(n):
            return 0 if (grad or n > 64) else (1 if n > 32 else 2)
        count = {}
        for sn in objs:
            c = small_class(len(self.y[sn]))
            count[c] = count.get(c, 0) + 1
        for sn in objs:
            n = len(self.y[sn])
            # (objects of at most 32 / 64 points get buckets of their own when there are enough of them: the library fits such a
            # batch four / two objects per 128 x 128 tile, which it can only do when EVERY object of the call is that short)
            small = small_class(n)
            if count.get(small, 0) < 8:
                small = 0
            buckets.setdefault(((n + 127) // 128 * 128, -small), []).append(sn)
        chunks = []
        for bkey in sorted(buckets, reverse=True):
            np_ = bkey[0]
            rows = (3 * np_ + 128) if grad else (np_ + (int(m) + 1 + 127) // 128 * 128)
            per = max(1, min(MAX_BATCH_OBJECTS, int(BATCH_BYTES_LIMIT // (rows * np_ * 8))))
            members = buckets[bkey]
            for lo in range(0, len(members), per):
                chunks.append(members[lo:lo + per])
        self._chunks_cache = (key, chunks)
        return chunks

    def _fit_single(self, sn, kp, nugget, grid=None, new_y0=None, kss_noise=0.):
        """One cgp_fit for object sn; its coordinates / residual are uploaded only when another
        object (or another GP on the same handle) replaced them."""
        dev = self.device
        diag = self._make_resident(sn)    # None keeps the resident diagonal vector
        self._factor_token = None
        out = dev.fit(kp, diag, grid, new_y0, kss_noise)
        self._factor_token = self._res_token(sn) + (kp.tobytes(), float(nugget))
        return out

    def _ensure_factor(self, sn, kp, nugget):
        """Make object sn's factor (for kp, nugget) the one in HBM; re-fit if it was replaced."""
        tok = self._res_token(sn)
        if self._resident == tok and self._factor_token == tok + (kp.tobytes(), float(nugget)):
            return
        self._fit_single(sn, kp, nugget)

    def _fit_single_svd(self, sn, kp, grid=None, new_y0=None, kss_noise=0., want_cov=False):
        """One cgp_fit_svd (the reference's default svd_method=True algorithm) for object sn."""
        dev = self.device
        diag = self._make_resident(sn)
        self._factor_token = None              # the handle's Cholesky factor does not survive this call
        return dev.fit_svd(kp, diag, grid, new_y0, kss_noise, SVD_THRESHOLD, want_cov)

    # ------------------------------------------------------------------ likelihood
    def compute_log_likelihood(self, Hyperparameter, svd_method=True):
        """Summed log marginal likelihood of all objects for one hyperparameter set
        (Gaussian_process.py:191-213); result in self.log_likelihood, shape (1,)."""
        if self.fit_nugget:
            Nugget = Hyperparameter[-1]
            hyperparameter = Hyperparameter[:-1]
        else:
            Nugget = self.nugget
            hyperparameter = Hyperparameter
        objs = self._my_objects()
        per_obj = None
        if not svd_method or self._svd_by_cholesky(Nugget):
            # Cholesky route; every rank learns whether ANY rank met a non-positive pivot before the gather, so
            # that all of them raise (svd_method=False) or fall back to the eigensolver route together
            err = None
            try:
                per_obj = self._loglik_chol(objs, hyperparameter, Nugget)
            except np.linalg.LinAlgError as e:
                err = e
            failed = err is not None if self.comm is None else self.comm.agree_any(err is not None)
            if failed:
                per_obj = None
                if not svd_method:
                    raise err if err is not None else np.linalg.LinAlgError(
                        'the matrix of an object on another rank is not positive definite')
        if per_obj is None:
            kp = _kparams(self._dim, hyperparameter, Nugget)
            per_obj = [self._fit_single_svd(sn, kp)[0] for sn in objs]
        if self.comm is not None:
            per_obj = self.comm.gather_units(np.asarray(per_obj, dtype=np.float64), self.N_sn)
        total = 0.                              # object order, like the reference's loop (the same float64 additions)
        for v in per_obj:
            total = total + float(v)
        self.log_likelihood = np.array([total]) if len(per_obj) else 0

    def _batch_upload(self, chunk):
        """Make `chunk` the resident batch of the handle (skipped when it already is)."""
        dev = self.device

        def token():
            if len(chunk) == self.N_sn:         # every object, in order: the ids of the three lists' entries at C speed
                ids = (tuple(map(id, self.y)), tuple(map(id, self.Time)), tuple(map(id, self.y_err)))
            else:
                ids = tuple((id(self.y[sn]), id(self.Time[sn]), id(self.y_err[sn])) for sn in chunk)
            return (dev.batch_generation, tuple(chunk), ids)
        if getattr(self, '_batch_resident', None) != token():
            dev.batch_set_data(self._dim, [self.Time[sn] for sn in chunk], [self._resid(sn) for sn in chunk],
                               [self._diag(sn) for sn in chunk])
            self._batch_resident = token()

    def _loglik_chol(self, objs, hyperparameter, nugget):
        dev = self.device
        kp = _kparams(self._dim, hyperparameter, nugget)
        if len(objs) == 1:
            return [self._fit_single(objs[0], kp, nugget)[0]]
        where = {sn: k for k, sn in enumerate(objs)}
        out = [0.] * len(objs)
        for chunk in self._batch_chunks(objs):
            self._batch_upload(chunk)      # objects stay in HBM across objective evaluations (single-chunk case)
            ll, info, _, _ = dev.batch_fit_resident(kp)
            bad = np.nonzero(info)[0]
            if len(bad):
                raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite'
                                            ' (object %d)' % (info[bad[0]], chunk[bad[0]]))
            for c, sn in enumerate(chunk):
                out[where[sn]] = float(ll[c])
        return out

    def compute_log_likelihood_gradient(self, Hyperparameter):
        """Log-likelihood (as compute_log_likelihood, Cholesky path) AND its analytic gradient with
        respect to the entries of `Hyperparameter` (trailing nugget included when self.fit_nugget):
            d logL / d theta = 1/2 tr((alpha alpha' - K^-1) dK/dtheta),   alpha = K^-1 (y - y0).
        Extension over the reference, which only has the derivative-free
        search of Gaussian_process.py:247.  Per object one device call (cgp_fit_grad): factor,
        alpha, K^-1 and the reduction against the kernel derivatives all stay in HBM; five moment
        sums come back and kernel.gradient_from_sums_* applies the chain rule.
        Sets self.log_likelihood (shape (1,)) and self.log_likelihood_gradient; returns the latter."""
        if self.fit_nugget:
            Nugget = Hyperparameter[-1]
            hyperparameter = Hyperparameter[:-1]
        else:
            Nugget = self.nugget
            hyperparameter = Hyperparameter
        nh = len(hyperparameter)
        kp = _kparams(self._dim, hyperparameter, Nugget)
        dev = self.device
        objs = self._my_objects()
        rows = np.zeros((len(objs), 2 + nh))
        from_sums = _kernel.gradient_from_sums_1d if self._dim == 1 else _kernel.gradient_from_sums_2d
        if len(objs) > 1:
            # objects in lock-step: one launch sequence per chunk of the batch (cgp_batch_fit_grad)
            where = {sn: k for k, sn in enumerate(objs)}
            for chunk in self._batch_chunks(objs, grad=True):
                self._batch_upload(chunk)
                ll, info, sums = dev.batch_fit_grad_resident(kp)
                bad = np.nonzero(info)[0]
                if len(bad):
                    raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite'
                                                ' (object %d)' % (info[bad[0]], chunk[bad[0]]))
                for c, sn in enumerate(chunk):
                    g = from_sums(hyperparameter, Nugget, sums[c])
                    pos = where[sn]
                    rows[pos, 0] = ll[c]
                    rows[pos, 1:1 + nh] = g[:nh]
                    rows[pos, 1 + nh] = g[-1]
            objs = []
        for k, sn in enumerate(objs):
            diag = self._make_resident(sn)
            self._factor_token = None
            ll, _, sums = dev.fit_grad(kp, diag)
            self._factor_token = self._res_token(sn) + (kp.tobytes(), float(Nugget))
            if self._dim == 1:
                g = _kernel.gradient_from_sums_1d(hyperparameter, Nugget, sums)
            else:
                g = _kernel.gradient_from_sums_2d(hyperparameter, Nugget, sums)
            rows[k, 0] = ll
            rows[k, 1:1 + nh] = g[:nh]
            rows[k, 1 + nh] = g[-1]
        if self.comm is not None:
            rows = self.comm.gather_units(rows, self.N_sn)
        log_likelihood = 0
        for v in rows[:, 0]:
            log_likelihood = log_likelihood + np.array([v])
        self.log_likelihood = log_likelihood
        grad = rows[:, 1:].sum(axis=0)
        self.log_likelihood_gradient = grad if self.fit_nugget else grad[:nh]
        return self.log_likelihood_gradient

    def find_hyperparameters(self, hyperparameter_guess=None, nugget=False, svd_method=True, method=None):
        """Maximum-likelihood hyperparameters.  method=None: Nelder-Mead (scipy.optimize.fmin), the
        reference's driver (Gaussian_process.py:216-253); every evaluation of the objective is
        one device call with the data already resident in HBM.  method='L-BFGS-B' (or any
        scipy.optimize.minimize method that takes a gradient): opt-in gradient-based search on the
        Cholesky path with the analytic gradient of compute_log_likelihood_gradient (extension;
        svd_method is ignored there)."""
        if hyperparameter_guess is not None:
            assert len(self.hyperparameters) == len(hyperparameter_guess), 'should be same len'
            self.hyperparameters = hyperparameter_guess

        def cost(hyper):
            self.compute_log_likelihood(hyper, svd_method=svd_method)
            return -self.log_likelihood[0]

        start = [self.hyperparameters[i] for i in range(len(self.hyperparameters))]
        self.fit_nugget = bool(nugget)
        if self.fit_nugget:
            start.append(1.)
        if method is None:
            best = fmin(cost, start, disp=False)
        else:
            from scipy.optimize import minimize

            def cost_grad(hyper):
                g = self.compute_log_likelihood_gradient(hyper)
                return -self.log_likelihood[0], -np.asarray(g, dtype=np.float64)

            res = minimize(cost_grad, np.asarray(start, dtype=np.float64), jac=True, method=method)
            self.fit_result = res
            best = res.x
        self.hyperparameters = np.array(self.hyperparameters, dtype=np.float64)
        for i in range(len(self.hyperparameters)):
            self.hyperparameters[i] = np.sqrt(best[i] ** 2)
        if self.fit_nugget:
            self.nugget = np.sqrt(best[-1] ** 2)

    # ------------------------------------------------------------------ matrices
    def compute_kernel_matrix(self):
        """self.kernel_matrix = [K_sn] (Gaussian_process.py:256-267), produced lazily on the
        device with the hyperparameters current at call time."""
        hyp = np.array(self.hyperparameters, dtype=np.float64)
        nugget = self.nugget

        def make(sn):
            return self.kernel(self.Time[sn], hyp, nugget=nugget, y_err=self.y_err[sn], device=self.device)

        self.kernel_matrix = _LazyList(self.N_sn, make)

    # ------------------------------------------------------------------ prediction
    def _grid(self, sn):
        if not self.as_the_same_time:
            return self.new_binning
        if REFERENCE_STALE_INDEX:
            return self.new_binning[self.N_sn - 1]
        return self.new_binning[sn]

    def _new_y0(self, sn):
        if self.substract_mean or self.Mean_Y is not None:
            if self.as_the_same_time:
                return self.y0[sn]
            return return_mean(self.y[sn], self.Time[sn], new_x=self.new_binning, mean_y=self.Mean_Y,
                               mean_xaxis=self.Time_mean, diff=self.diff[sn])
        return 0.

    def get_prediction(self, new_binning=None, COV=True, svd_method=True):
        """Interpolate every object on new_binning (None: on its own grid) --
        Gaussian_process.py:270-338.  Results: self.Prediction (list of (M,)),
        self.variance (list of (M,), the diagonal of the predictive covariance; extension),
        and with COV=True self.covariance_matrix (lazy list of (M, M))."""
        if new_binning is None:
            self.as_the_same_time = True
            self.new_binning = self.Time
        else:
            self.as_the_same_time = False
            self.new_binning = new_binning
        self.compute_kernel_matrix()
        hyp = np.array(self.hyperparameters, dtype=np.float64)
        nugget = self.nugget
        done = False
        if not svd_method or self._svd_by_cholesky(nugget):
            done = self._predict_chol(hyp, nugget, must_succeed=not svd_method)
        if not done:
            self._predict_svd(hyp, nugget, COV)
        if COV:
            self.get_covariance_matrix()

    def _predict_chol(self, hyp, nugget, must_succeed):
        """Cholesky route of get_prediction.  Returns False (nothing stored) when some object, on ANY rank, is not
        positive definite and must_succeed is False; raises numpy.linalg.LinAlgError on every rank otherwise."""
        dev = self.device
        kp = _kparams(self._dim, hyp, nugget)
        kss_noise = 0. ** 2 + 0. ** 2 + nugget ** 2          # kernel(new_grid, hyp, nugget=...) :357
        objs = self._my_objects()
        shared = not self.as_the_same_time
        pred = [None] * self.N_sn
        variance = [None] * self.N_sn
        err = None
        try:
            if len(objs) > 1 and shared:
                m = len(self.new_binning)
                for chunk in self._batch_chunks(objs, m=m):
                    self._batch_upload(chunk)
                    _, info, mu, var = dev.batch_fit_resident(kp, xnew=self.new_binning, kss_noise=kss_noise)
                    bad = np.nonzero(info)[0]
                    if len(bad):
                        raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite'
                                                    ' (object %d)' % (info[bad[0]], chunk[bad[0]]))
                    for k, sn in enumerate(chunk):
                        new_y0 = self._new_y0(sn)
                        self.warning_pf = new_y0
                        pred[sn] = mu[k] + new_y0
                        variance[sn] = var[k]
            elif len(objs) > 1:
                # each object on its own grid, still ONE launch sequence per chunk: grids padded to the longest
                # (extra points repeat the last one and are dropped)
                m_all = max(len(self._grid(sn)) for sn in objs)
                for chunk in self._batch_chunks(objs, m=m_all):
                    self._batch_upload(chunk)
                    grids = [np.asarray(self._grid(sn), dtype=np.float64) for sn in chunk]
                    m_max = max(len(g) for g in grids)
                    padded = np.stack([np.concatenate([g, np.repeat(g[-1:], m_max - len(g), axis=0)]) for g in grids])
                    _, info, mu, var = dev.batch_fit_resident(kp, xnew=padded, kss_noise=kss_noise,
                                                              xnew_per_object=True)
                    bad = np.nonzero(info)[0]
                    if len(bad):
                        raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite'
                                                    ' (object %d)' % (info[bad[0]], chunk[bad[0]]))
                    for k, sn in enumerate(chunk):
                        new_y0 = self._new_y0(sn)
                        self.warning_pf = new_y0
                        pred[sn] = mu[k, :len(grids[k])] + new_y0
                        variance[sn] = var[k, :len(grids[k])].copy()
            else:
                for sn in objs:
                    new_y0 = self._new_y0(sn)
                    self.warning_pf = new_y0
                    _, _, mu, var = self._fit_single(sn, kp, nugget, self._grid(sn), new_y0, kss_noise)
                    pred[sn] = mu
                    variance[sn] = var
        except np.linalg.LinAlgError as e:
            err = e
        # agreement BEFORE the gathers: a rank that raised must not leave its peers inside a collective
        failed = err is not None if self.comm is None else self.comm.agree_any(err is not None)
        if failed:
            if must_succeed:
                raise err if err is not None else np.linalg.LinAlgError(
                    'the matrix of an object on another rank is not positive definite')
            return False
        if self.comm is not None:
            pred = self.comm.gather_unit_arrays(pred, self.N_sn)
            variance = self.comm.gather_unit_arrays(variance, self.N_sn)
        self.Prediction = pred
        self.variance = variance

        def make_inv(sn):
            self._ensure_factor(sn, kp, nugget)
            return dev.fetch(_lib.MAT_KINV)

        self.inv_kernel_matrix = _LazyList(self.N_sn, make_inv)
        # recorded whatever COV says: the reference allows get_prediction(COV=False); get_covariance_matrix()
        self._cov_state = ('chol', kp, nugget, kss_noise, {})
        return True

    def get_covariance_matrix(self):
        """self.covariance_matrix = [K(new,new; nugget) - H K^-1 H^T] (Gaussian_process.py:340-361) for the
        hyperparameters, nugget, grid and route of the LAST get_prediction.  The full (M, M) matrix is built on
        the device when an item is first read (Cholesky route: K** - W W^T, one MFMA GEMM on the resident factor;
        eigensolver route: K** - T H' by cgp_fit_svd, kept when get_prediction already produced it)."""
        state = getattr(self, '_cov_state', None)
        if state is None:
            raise AttributeError('get_covariance_matrix needs get_prediction first')
        route, kp, nugget, kss_noise, ready = state
        dev = self.device

        def make(sn):
            grid = self.new_binning[sn] if self.as_the_same_time else self.new_binning
            if route == 'chol':
                self._ensure_factor(sn, kp, nugget)
                _, _, cov = dev.predict(grid, None, kss_noise, want_cov=True)
                return cov
            if sn in ready:
                return ready[sn]
            return self._fit_single_svd(sn, kp, grid, None, kss_noise, want_cov=True)[5]

        self.covariance_matrix = _LazyList(self.N_sn, make)

    def _predict_svd(self, hyp, nugget, COV=True):
        """Reference default (svd_method=True): pseudo-inverse route, one device call per object
        (cgp_fit_svd: K, its eigen-decomposition, K^+, H K^+ r and K** - H K^+ H' stay in HBM).  With a
        communicator every rank serves its own objects and the O(M) results are gathered."""
        kp = _kparams(self._dim, hyp, nugget)
        kss_noise = float(nugget) ** 2
        hyp = np.array(hyp, dtype=np.float64)
        covs = {}
        pred = [None] * self.N_sn
        variance = [None] * self.N_sn
        for sn in self._my_objects():
            new_y0 = self._new_y0(sn)
            self.warning_pf = new_y0
            grid = self._grid(sn)
            cgrid = self.new_binning[sn] if self.as_the_same_time else self.new_binning
            same = grid is cgrid or np.array_equal(grid, cgrid)
            _, _, _, mu, var, cov = self._fit_single_svd(sn, kp, grid, new_y0, kss_noise, want_cov=COV and same)
            pred[sn] = mu
            if not same:
                # REFERENCE_STALE_INDEX: the mean uses the stale grid, the covariance the object's own (:351-354)
                _, _, _, _, var, cov = self._fit_single_svd(sn, kp, cgrid, None, kss_noise, want_cov=COV)
            if cov is not None:
                covs[sn] = cov
            variance[sn] = var
        if self.comm is not None:
            pred = self.comm.gather_unit_arrays(pred, self.N_sn)
            variance = self.comm.gather_unit_arrays(variance, self.N_sn)
        self.Prediction = pred
        self.variance = variance
        self._cov_state = ('svd', kp, nugget, kss_noise, covs)

        def make_inv(sn):
            return svd_inverse(self.kernel(self.Time[sn], hyp, nugget=nugget, y_err=self.y_err[sn], device=self.device),
                               device=self.device)

        self.inv_kernel_matrix = _LazyList(self.N_sn, make_inv)


class gaussian_process(Gaussian_process):
    """One object (mirror of cosmogp.gaussian_process, Gaussian_process.py:365-379)."""

    def __init__(self, y, Time, kernel='RBF1D', y_err=None, diff=None, Mean_Y=None,
                 Time_mean=None, substract_mean=False, device=None):
        if y_err is not None:
            y_err = [y_err]
        Gaussian_process.__init__(self, [y], [Time], kernel=kernel, y_err=y_err, Mean_Y=Mean_Y,
                                  Time_mean=Time_mean, diff=diff, substract_mean=substract_mean,
                                  device=device)


class gaussian_process_nobject(Gaussian_process):
    """n objects sharing hyperparameters (mirror of cosmogp.gaussian_process_nobject,
    Gaussian_process.py:382-393)."""

    def __init__(self, y, Time, kernel='RBF1D', y_err=None, diff=None, Mean_Y=None,
                 Time_mean=None, substract_mean=False, device=None):
        Gaussian_process.__init__(self, y, Time, kernel=kernel, y_err=y_err, diff=diff, Mean_Y=Mean_Y,
                                  Time_mean=Time_mean, substract_mean=substract_mean, device=device)


def gaussian_process_2d(y, Time, **kwargs):
    """Convenience constructor named in BASELINE.json's north_star (no such symbol exists in
    the reference): gaussian_process(..., kernel='RBF2D')."""
    kwargs['kernel'] = 'RBF2D'
    return gaussian_process(y, Time, **kwargs)
