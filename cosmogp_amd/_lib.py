"""ctypes binding of libcosmogp_hip.so (include/cosmogp_hip.h).

Thin by design: numpy arrays in, numpy arrays out, status codes turned into the
exceptions the reference raises (numpy.linalg.LinAlgError for a non positive
definite matrix, cosmogp/inv_matrix.py:23).  There is NO CPU fallback: if the
shared library or a GPU is missing every entry point raises HipUnavailable.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcosmogp_hip.so')

MAT_K, MAT_L, MAT_KINV = 0, 1, 2

#: every symbol include/cosmogp_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    'cgp_version', 'cgp_device_count', 'cgp_create', 'cgp_destroy', 'cgp_last_error',
    'cgp_synchronize', 'cgp_rbf', 'cgp_cholesky_inverse', 'cgp_svd_inverse', 'cgp_eig_info', 'cgp_fit_set_data', 'cgp_fit', 'cgp_fit_svd',
    'cgp_fit_predict', 'cgp_fit_grad', 'cgp_fit_fetch', 'cgp_batch_set_data', 'cgp_batch_fit', 'cgp_batch_fit_grad', 'cgp_batch_loo', 'cgp_sweep_loglik',
    'cgp_comm_unique_id', 'cgp_comm_init', 'cgp_comm_destroy', 'cgp_comm_count', 'cgp_comm_allgather', 'cgp_comm_allreduce',
    'cgp_sweep_loglik_sharded', 'cgp_bispev', 'cgp_bench_fit', 'cgp_bench_fit_objects', 'cgp_microbench',
    'cgp_selftest_mfma',
)
COMM_ID_BYTES = 128
OP_SUM, OP_MAX = 0, 1


class HipUnavailable(RuntimeError):
    """libcosmogp_hip.so or an MI355X is missing; the product path has no CPU fallback."""


class HipError(RuntimeError):
    pass


_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64
_lib = None
_lib_lock = threading.RLock()


def load_library():
    """Load the shared library (works without a GPU; compute calls then fail loudly)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable('%s not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                                 'or `make -C cosmogp_amd/csrc`' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        lib.cgp_version.restype = ctypes.c_int
        lib.cgp_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
        lib.cgp_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        lib.cgp_destroy.argtypes = [vp]
        lib.cgp_last_error.argtypes = [vp]
        lib.cgp_last_error.restype = ctypes.c_char_p
        lib.cgp_synchronize.argtypes = [vp]
        lib.cgp_rbf.argtypes = [vp, ctypes.c_int, _dp, _i64, _dp, _i64, _dp, _dp, _dp, _i64]
        lib.cgp_cholesky_inverse.argtypes = [vp, _dp, _i64, _i64, _dp, _dp]
        lib.cgp_svd_inverse.argtypes = [vp, _dp, _i64, _i64, ctypes.c_double, _dp, _dp, ctypes.POINTER(_i64)]
        lib.cgp_fit_svd.argtypes = [vp, _dp, _dp, _dp, _i64, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp,
                                    ctypes.POINTER(_i64), _dp, _dp, _dp]
        lib.cgp_fit_set_data.argtypes = [vp, ctypes.c_int, _dp, _dp, _i64]
        lib.cgp_fit.argtypes = [vp, _dp, _dp, _dp, _i64, _dp, ctypes.c_double, _dp, _dp, _dp, _dp]
        lib.cgp_fit_predict.argtypes = [vp, _dp, _i64, _dp, ctypes.c_double, _dp, _dp, _dp]
        lib.cgp_fit_grad.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
        lib.cgp_fit_fetch.argtypes = [vp, ctypes.c_int, _dp]
        lib.cgp_batch_set_data.argtypes = [vp, ctypes.c_int, _i64, ctypes.POINTER(_i64), _dp, _dp, _dp]
        lib.cgp_batch_fit.argtypes = [vp, ctypes.c_int, _i64, ctypes.POINTER(_i64), _dp, _dp, _dp, _dp, _dp,
                                      _i64, ctypes.c_int, _dp, ctypes.c_double, _dp,
                                      ctypes.POINTER(ctypes.c_int32), _dp, _dp]
        lib.cgp_batch_fit_grad.argtypes = [vp, _dp, _dp, ctypes.POINTER(ctypes.c_int32), _dp]
        lib.cgp_batch_loo.argtypes = [vp, _dp, _dp, _dp, _dp, ctypes.POINTER(ctypes.c_int32)]
        lib.cgp_sweep_loglik.argtypes = [vp, _i64, _dp, _dp, _dp, ctypes.POINTER(ctypes.c_int32)]
        lib.cgp_comm_unique_id.argtypes = [ctypes.c_char_p]
        lib.cgp_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        lib.cgp_comm_destroy.argtypes = [vp]
        lib.cgp_comm_count.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        lib.cgp_eig_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
        lib.cgp_comm_allgather.argtypes = [vp, _dp, _i64, _dp]
        lib.cgp_comm_allreduce.argtypes = [vp, _dp, _i64, ctypes.c_int]
        lib.cgp_sweep_loglik_sharded.argtypes = [vp, _i64, _dp, _dp, _i64, _dp, ctypes.POINTER(ctypes.c_int32)]
        lib.cgp_bispev.argtypes = [vp, _dp, _i64, _dp, _i64, _dp, ctypes.c_int, ctypes.c_int, _dp, _i64, _dp]
        lib.cgp_bench_fit.argtypes = [vp, ctypes.c_int, _dp, _dp]
        lib.cgp_bench_fit_objects.argtypes = [vp, ctypes.c_int, _i64, _dp]
        lib.cgp_microbench.argtypes = [vp, _dp, _dp]
        lib.cgp_selftest_mfma.argtypes = [vp]
        _lib = lib
        return lib


def comm_unique_id():
    """128 opaque bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
    lib = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    st = lib.cgp_comm_unique_id(buf)
    if st != 0:
        raise HipUnavailable('cgp_comm_unique_id failed (status %d): librccl.so.1 missing?' % st)
    return buf.raw


def device_count():
    lib = load_library()
    c = ctypes.c_int(0)
    lib.cgp_device_count(ctypes.byref(c))
    return c.value


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError('expected shape %r, got %r' % (shape, a.shape))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Device(object):
    """One handle = one GPU + one HIP stream + its HBM workspaces.  Not thread safe."""

    def __init__(self, index=None):
        lib = load_library()
        if index is None:
            index = int(os.environ.get('COSMOGP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        n = device_count()
        if n == 0:
            raise HipUnavailable('no HIP device visible: cosmogp_amd needs an MI355X (gfx950); '
                                 'there is no CPU fallback')
        self.index = index % n
        self._lib = lib
        self._h = ctypes.c_void_p()
        st = lib.cgp_create(self.index, ctypes.byref(self._h))
        if st != 0:
            raise HipUnavailable('cgp_create(device=%d) failed with status %d' % (self.index, st))
        self.dim = None
        self.n = None
        #: bumped whenever the resident single-object factor is replaced (lazy attributes
        #: of Gaussian_process check it before fetching from the device)
        self.generation = 0
        self.batch_generation = 0       # same for the resident batch
        self._batch = None
        self.comm_rank, self.comm_world = 0, 1

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.cgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- status handling ---------------------------------------------------------
    def _check(self, st, what):
        if st == 0:
            return
        if st > 0:
            raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite' % st)
        msg = self._lib.cgp_last_error(self._h)
        raise HipError('%s failed (status %d): %s' % (what, st, msg.decode() if msg else ''))

    # ---- operator level ------------------------------------------------------------
    def rbf(self, dim, x, kp, xnew=None, diag=None):
        x = _f64(x)
        n = x.shape[0]
        kp = _f64(kp, (7,))
        if xnew is None:
            out = np.empty((n, n))
            dg = None if diag is None else _f64(diag, (n,))
            st = self._lib.cgp_rbf(self._h, dim, _ptr(x), n, None, 0, _ptr(kp), _ptr(dg), _ptr(out), n)
        else:
            xnew = _f64(xnew)
            m = xnew.shape[0]
            out = np.empty((m, n))
            st = self._lib.cgp_rbf(self._h, dim, _ptr(x), n, _ptr(xnew), m, _ptr(kp), None, _ptr(out), n)
        self._check(st, 'cgp_rbf')
        return out

    def cholesky_inverse(self, K):
        K = _f64(K)
        n = K.shape[0]
        if K.ndim != 2 or K.shape[1] != n:
            raise ValueError('expected a square matrix')
        inv = np.empty((n, n))
        ld = ctypes.c_double(0.)
        self.generation += 1
        st = self._lib.cgp_cholesky_inverse(self._h, _ptr(K), n, n, _ptr(inv), ctypes.byref(ld))
        self._check(st, 'cgp_cholesky_inverse')
        return inv, ld.value

    def svd_inverse(self, K, thresh=1e-15):
        """(pseudo-inverse, logdet, n_dropped) of a symmetric matrix by the device eigensolver."""
        K = _f64(K)
        n = K.shape[0]
        if K.ndim != 2 or K.shape[1] != n:
            raise ValueError('expected a square matrix')
        inv = np.empty((n, n))
        ld, nd = ctypes.c_double(0.), _i64(0)
        st = self._lib.cgp_svd_inverse(self._h, _ptr(K), n, n, float(thresh), _ptr(inv), ctypes.byref(ld),
                                       ctypes.byref(nd))
        self._check(st, 'cgp_svd_inverse')
        return inv, ld.value, nd.value

    # ---- fused level ---------------------------------------------------------------
    def set_data(self, dim, x, resid):
        x = _f64(x)
        n = x.shape[0]
        resid = _f64(resid, (n,))
        self.generation += 1
        self._check(self._lib.cgp_fit_set_data(self._h, dim, _ptr(x), _ptr(resid), n), 'cgp_fit_set_data')
        self.dim, self.n = dim, n

    def fit(self, kp, diag=None, xnew=None, mean_new=None, kss_noise=0., want_var=True):
        """Returns (loglik, logdet, mu, var); mu/var None when xnew is None.
        diag: uploaded and kept resident; None keeps the resident vector (zeros after set_data);
        kp[6] is added to every element of it on the device."""
        kp = _f64(kp, (7,))
        dg = None if diag is None else _f64(diag, (self.n,))
        ll, ld = ctypes.c_double(0.), ctypes.c_double(0.)
        if xnew is None:
            st = self._lib.cgp_fit(self._h, _ptr(kp), _ptr(dg), None, 0, None, float(kss_noise),
                                   ctypes.byref(ll), ctypes.byref(ld), None, None)
            self._check(st, 'cgp_fit')
            return ll.value, ld.value, None, None
        xnew = _f64(xnew)
        m = xnew.shape[0]
        mn = None if mean_new is None else _f64(np.broadcast_to(mean_new, (m,)))
        mu = np.empty(m)
        var = np.empty(m) if want_var else None
        st = self._lib.cgp_fit(self._h, _ptr(kp), _ptr(dg), _ptr(xnew), m, _ptr(mn), float(kss_noise),
                               ctypes.byref(ll), ctypes.byref(ld), _ptr(mu), _ptr(var))
        self._check(st, 'cgp_fit')
        return ll.value, ld.value, mu, var

    def fit_svd(self, kp, diag=None, xnew=None, mean_new=None, kss_noise=0., thresh=1e-15, want_cov=False):
        """The fit by the reference's svd_method=True algorithm, on the device (cgp_fit_svd).
        Returns (loglik, logdet, n_dropped, mu, var, cov)."""
        kp = _f64(kp, (7,))
        dg = None if diag is None else _f64(diag, (self.n,))
        ll, ld, nd = ctypes.c_double(0.), ctypes.c_double(0.), _i64(0)
        if xnew is None:
            st = self._lib.cgp_fit_svd(self._h, _ptr(kp), _ptr(dg), None, 0, None, float(kss_noise), float(thresh),
                                       ctypes.byref(ll), ctypes.byref(ld), ctypes.byref(nd), None, None, None)
            self._check(st, 'cgp_fit_svd')
            return ll.value, ld.value, nd.value, None, None, None
        xnew = _f64(xnew)
        m = xnew.shape[0]
        mn = None if mean_new is None else _f64(np.broadcast_to(mean_new, (m,)))
        mu, var = np.empty(m), np.empty(m)
        cov = np.empty((m, m)) if want_cov else None
        st = self._lib.cgp_fit_svd(self._h, _ptr(kp), _ptr(dg), _ptr(xnew), m, _ptr(mn), float(kss_noise),
                                   float(thresh), ctypes.byref(ll), ctypes.byref(ld), ctypes.byref(nd), _ptr(mu),
                                   _ptr(var), _ptr(cov))
        self._check(st, 'cgp_fit_svd')
        return ll.value, ld.value, nd.value, mu, var, cov

    def fit_grad(self, kp, diag=None):
        """Returns (loglik, logdet, sums[5]): the log-likelihood of the resident object and the moment
        sums of its analytic gradient (include/cosmogp_hip.h, cgp_fit_grad)."""
        kp = _f64(kp, (7,))
        dg = None if diag is None else _f64(diag, (self.n,))
        ll, ld = ctypes.c_double(0.), ctypes.c_double(0.)
        sums = np.zeros(5)
        st = self._lib.cgp_fit_grad(self._h, _ptr(kp), _ptr(dg), ctypes.byref(ll), ctypes.byref(ld), _ptr(sums))
        self._check(st, 'cgp_fit_grad')
        return ll.value, ld.value, sums

    def predict(self, xnew, mean_new=None, kss_noise=0., want_cov=False):
        xnew = _f64(xnew)
        m = xnew.shape[0]
        mn = None if mean_new is None else _f64(np.broadcast_to(mean_new, (m,)))
        mu, var = np.empty(m), np.empty(m)
        cov = np.empty((m, m)) if want_cov else None
        st = self._lib.cgp_fit_predict(self._h, _ptr(xnew), m, _ptr(mn), float(kss_noise), _ptr(mu), _ptr(var),
                                       _ptr(cov))
        self._check(st, 'cgp_fit_predict')
        return mu, var, cov

    def fetch(self, which):
        out = np.empty((self.n, self.n))
        self._check(self._lib.cgp_fit_fetch(self._h, which, _ptr(out)), 'cgp_fit_fetch')
        return out

    # ---- batched -------------------------------------------------------------------
    @staticmethod
    def _batch_host_arrays(xs, resids, diags):
        """(x, resid, diag, ns) as the flat float64 arrays the C ABI takes"""
        B = len(xs)
        packed = (isinstance(xs, np.ndarray) and isinstance(resids, np.ndarray) and resids.ndim == 2 and xs.shape[:2] == resids.shape
                  and (diags is None or (isinstance(diags, np.ndarray) and diags.shape == resids.shape)))
        if packed:
            # ONE (B, n[, 2]) array per quantity (objects of equal length): no per-object work on the host at all
            ns = np.full(B, resids.shape[1], dtype=np.int64)
            x, r = _f64(xs).ravel(), _f64(resids).ravel()
            d = None if diags is None else _f64(diags).ravel()
        else:
            ns = np.array([len(r) for r in resids], dtype=np.int64)
            x = _f64(np.concatenate([np.asarray(a, dtype=np.float64).reshape(len(a), -1) for a in xs]).ravel())
            r = _f64(np.concatenate(resids))
            d = None if diags is None else _f64(np.concatenate(diags))
        return x, r, d, ns

    def batch_set_data(self, dim, xs, resids, diags=None):
        """Upload B objects and keep them resident: lists of per-object arrays (ragged allowed), or -- objects of equal
        length -- one C-contiguous (B, n[, 2]) / (B, n) / (B, n) array per quantity, which goes down without any per-object
        work on the host."""
        x, r, d, ns = self._batch_host_arrays(xs, resids, diags)
        B = len(ns)
        self.batch_generation += 1
        st = self._lib.cgp_batch_set_data(self._h, dim, B, ns.ctypes.data_as(ctypes.POINTER(_i64)), _ptr(x), _ptr(r),
                                          _ptr(d))
        self._check(st, 'cgp_batch_set_data')
        self._batch = (dim, B)
        self._batch_nmax = int(ns.max()) if B else 0

    def batch_fit_resident(self, kp, xnew=None, mean_new=None, kss_noise=0., xnew_per_object=False):
        """Fit the resident batch: only kp (7 doubles) and the prediction grid go down.
        Returns (loglik[B], info[B], mu[B,m] or None, var[B,m] or None)."""
        dim, B = self._batch
        kp = _f64(kp, (7,))
        ll = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        if xnew is None:
            m, xn, mu, var, mn = 0, None, None, None, None
        else:
            xn = _f64(xnew)
            m = xn.shape[1] if xnew_per_object else xn.shape[0]
            if xnew_per_object and xn.shape[0] != B:
                raise ValueError('per-object grids need shape (B, m[, 2])')
            mu, var = np.empty((B, m)), np.empty((B, m))
            mn = None if mean_new is None else _f64(np.broadcast_to(mean_new, (m,)))
        st = self._lib.cgp_batch_fit(self._h, dim, B, None, None, None, None, _ptr(kp), _ptr(xn), m,
                                     1 if xnew_per_object else 0, _ptr(mn), float(kss_noise), _ptr(ll),
                                     info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(mu), _ptr(var))
        self._check(st, 'cgp_batch_fit')
        return ll, info, mu, var

    def batch_fit(self, dim, xs, resids, diags, kp, xnew=None, mean_new=None, kss_noise=0.,
                  xnew_per_object=False):
        """Upload + fit in one go (see batch_set_data / batch_fit_resident).  xnew: (m,[2]) grid shared by
        all objects, or (B, m,[2]) with xnew_per_object=True."""
        x, r, d, ns = self._batch_host_arrays(xs, resids, diags)
        B = len(ns)
        kp = _f64(kp, (7,))
        ll = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        if xnew is None:
            m, xn, mu, var, mn = 0, None, None, None, None
        else:
            xn = _f64(xnew)
            m = xn.shape[1] if xnew_per_object else xn.shape[0]
            if xnew_per_object and xn.shape[0] != B:
                raise ValueError('per-object grids need shape (B, m[, 2])')
            mu, var = np.empty((B, m)), np.empty((B, m))
            mn = None if mean_new is None else _f64(np.broadcast_to(mean_new, (m,)))
        self.batch_generation += 1
        # ONE call with the host data: from two chunks of objects on the library runs the upload of chunk k + 1 under the
        # device pass of chunk k (csrc/api.hip, batch_fit_pipelined)
        st = self._lib.cgp_batch_fit(self._h, dim, B, ns.ctypes.data_as(ctypes.POINTER(_i64)), _ptr(x), _ptr(r), _ptr(d),
                                     _ptr(kp), _ptr(xn), m, 1 if xnew_per_object else 0, _ptr(mn), float(kss_noise), _ptr(ll),
                                     info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(mu), _ptr(var))
        self._check(st, 'cgp_batch_fit')
        self._batch = (dim, B)
        self._batch_nmax = int(ns.max()) if B else 0
        return ll, info, mu, var

    def batch_fit_grad_resident(self, kp):
        """Log-likelihood and gradient moment sums of every object of the resident batch, one launch sequence.
        Returns (loglik[B], info[B], sums[B, 5])."""
        dim, B = self._batch
        kp = _f64(kp, (7,))
        ll = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        sums = np.zeros((B, 5))
        st = self._lib.cgp_batch_fit_grad(self._h, _ptr(kp), _ptr(ll), info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                          _ptr(sums))
        self._check(st, 'cgp_batch_fit_grad')
        return ll, info, sums

    def batch_loo(self, kp):
        """Leave-one-out ingredients of every object of the resident batch (cgp_batch_loo): alpha = K^-1 r, beta = K^-1 1,
        g = diag(K^-1) as (B, Np) arrays (entries beyond an object's n are undefined) and info[B]."""
        dim, B = self._batch
        kp = _f64(kp, (7,))
        npad = -(-max(1, self._batch_nmax) // 128) * 128
        alpha, beta, g = np.zeros((B, npad)), np.zeros((B, npad)), np.zeros((B, npad))
        info = np.zeros(B, dtype=np.int32)
        st = self._lib.cgp_batch_loo(self._h, _ptr(kp), _ptr(alpha), _ptr(beta), _ptr(g),
                                     info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        self._check(st, 'cgp_batch_loo')
        return alpha, beta, g, info

    def sweep_loglik(self, kps, diag=None):
        """Log-likelihood of the resident object for every row of kps (P, 7) in one launch sequence.
        Returns (loglik[P], info[P]).  Invalidates the resident batch of this handle."""
        kps = _f64(kps)
        if kps.ndim != 2 or kps.shape[1] != 7:
            raise ValueError('kps must have shape (P, 7)')
        dg = None if diag is None else _f64(diag, (self.n,))
        P = kps.shape[0]
        ll = np.empty(P)
        info = np.zeros(P, dtype=np.int32)
        self.batch_generation += 1
        st = self._lib.cgp_sweep_loglik(self._h, P, _ptr(kps), _ptr(dg), _ptr(ll),
                                        info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        self._check(st, 'cgp_sweep_loglik')
        return ll, info

    # ---- mean function (edge of the path) ---------------------------------------------
    def bispev(self, tck, points):
        """Bivariate B-spline `tck` = (tx, ty, c, kx, ky) of scipy.interpolate.bisplrep at points (m, 2): one launch
        instead of one bisplev call per point (cosmogp/mean.py:63-65)."""
        tx, ty, c, kx, ky = tck
        tx, ty, c = _f64(tx), _f64(ty), _f64(c)
        pts = _f64(points)
        if pts.ndim != 2 or pts.shape[1] != 2:
            raise ValueError('points must have shape (m, 2)')
        out = np.empty(pts.shape[0])
        st = self._lib.cgp_bispev(self._h, _ptr(tx), tx.size, _ptr(ty), ty.size, _ptr(c), int(kx), int(ky), _ptr(pts),
                                  pts.shape[0], _ptr(out))
        self._check(st, 'cgp_bispev')
        return out

    # ---- multi-GPU (RCCL inside the library) -----------------------------------------
    def comm_init(self, rank, world, uid):
        """Join the RCCL communicator identified by the 128-byte `uid` (collective over all ranks)."""
        if len(uid) != COMM_ID_BYTES:
            raise ValueError('the RCCL unique id has %d bytes' % COMM_ID_BYTES)
        self._check(self._lib.cgp_comm_init(self._h, int(rank), int(world), bytes(uid)), 'cgp_comm_init')
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        self._check(self._lib.cgp_comm_destroy(self._h), 'cgp_comm_destroy')

    def eig_info(self):
        """(route, rank, sweeps) of the last eigen-route call: route 0 = Jacobi sweeps, 1 = low-rank pre-stage, 2 = certified
        Cholesky (include/cosmogp_hip.h, cgp_eig_info)."""
        out = (ctypes.c_int64 * 3)()
        self._check(self._lib.cgp_eig_info(self._h, out), 'cgp_eig_info')
        return int(out[0]), int(out[1]), int(out[2])

    def comm_count(self):
        """Ranks the RCCL communicator of this handle reports (ncclCommCount)."""
        c = ctypes.c_int(0)
        self._check(self._lib.cgp_comm_count(self._h, ctypes.byref(c)), 'cgp_comm_count')
        return int(c.value)

    def comm_allgather(self, send):
        """send: float64 array of the same size on every rank; returns (world, send.size)."""
        send = _f64(send).ravel()
        recv = np.empty((self.comm_world, send.size))
        self._check(self._lib.cgp_comm_allgather(self._h, _ptr(send), send.size, _ptr(recv)), 'cgp_comm_allgather')
        return recv

    def comm_allreduce(self, buf, op=OP_SUM):
        buf = _f64(buf).ravel().copy()
        self._check(self._lib.cgp_comm_allreduce(self._h, _ptr(buf), buf.size, int(op)), 'cgp_comm_allreduce')
        return buf

    def comm_barrier(self):
        self._check(self._lib.cgp_comm_allreduce(self._h, None, 0, OP_SUM), 'cgp_comm_allreduce')

    def sweep_loglik_sharded(self, kps, diag=None, chunk=0):
        """cgp_sweep_loglik over the ranks of this handle's communicator: the FULL (P, 7) table on every rank,
        the complete (loglik[P], info[P]) back on every rank (one ncclAllGather, device to device)."""
        kps = _f64(kps)
        if kps.ndim != 2 or kps.shape[1] != 7:
            raise ValueError('kps must have shape (P, 7)')
        dg = None if diag is None else _f64(diag, (self.n,))
        P = kps.shape[0]
        ll = np.empty(P)
        info = np.zeros(P, dtype=np.int32)
        self.batch_generation += 1
        st = self._lib.cgp_sweep_loglik_sharded(self._h, P, _ptr(kps), _ptr(dg), int(chunk), _ptr(ll),
                                                info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        self._check(st, 'cgp_sweep_loglik_sharded')
        return ll, info

    # ---- measurement ---------------------------------------------------------------
    def bench_fit(self, iters, stages=True):
        ms = ctypes.c_double(0.)
        st8 = np.zeros(12) if stages else None
        self._check(self._lib.cgp_bench_fit(self._h, int(iters), ctypes.byref(ms), _ptr(st8)), 'cgp_bench_fit')
        return ms.value, st8

    def bench_fit_objects(self, iters, nobj):
        """iters device-resident passes over the first nobj objects of the resident batch; ms per pass."""
        ms = ctypes.c_double(0.)
        self._check(self._lib.cgp_bench_fit_objects(self._h, int(iters), int(nobj), ctypes.byref(ms)),
                    'cgp_bench_fit_objects')
        return ms.value

    def microbench(self):
        a, b = ctypes.c_double(0.), ctypes.c_double(0.)
        self._check(self._lib.cgp_microbench(self._h, ctypes.byref(a), ctypes.byref(b)), 'cgp_microbench')
        return a.value, b.value

    def selftest_mfma(self):
        st = self._lib.cgp_selftest_mfma(self._h)
        if st != 0:
            msg = self._lib.cgp_last_error(self._h)
            raise HipError('MFMA self-test failed (%d): %s' % (st, msg.decode() if msg else ''))
        return True


_default = {}


def default_device():
    """Process-wide handle on COSMOGP_DEVICE / LOCAL_RANK (created lazily, thread safe)."""
    key = os.getpid()
    with _lib_lock:
        if key not in _default:
            _default.clear()
            _default[key] = Device()
        return _default[key]


_pool = {}


def device_pool(n, index=None):
    """n handles on one GPU (created lazily, cached per process).  Independent fits issued through
    different handles run on different HIP streams with their own HBM buffers: with 3-4 in flight the
    latency-bound leaf chain of one fit runs in the gaps of the others' bulk updates (measured +35 % to
    +110 % fits/s)."""
    first = default_device() if index is None else None
    idx = first.index if first is not None else index
    key = (os.getpid(), idx)
    with _lib_lock:
        devs = _pool.setdefault(key, [first] if first is not None else [])
        while len(devs) < n:
            devs.append(Device(idx))
        return devs[:n]
