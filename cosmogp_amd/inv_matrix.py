"""Mirror of cosmogp/inv_matrix.py: the two callables of the reference's `chol` / `svd` seam
(cosmogp/Gaussian_process.py:6-9)."""
import numpy as np
from scipy import linalg

from . import _lib


def cholesky_inverse(matrix, return_logdet=False, device=None):
    """Explicit inverse (+ log-determinant) of an SPD matrix on the GPU -- same contract as
    cosmogp/inv_matrix.py:21-31: the input is not modified, a new dense (N, N) array is
    returned, numpy.linalg.LinAlgError is raised when the matrix is not positive definite.
    Device algorithm: blocked Cholesky (K3-K5), W = I L^-T by the same sweep, K^-1 = W W^T."""
    dev = device if device is not None else _lib.default_device()
    inv, logdet = dev.cholesky_inverse(np.asarray(matrix, dtype=np.float64))
    if return_logdet:
        return inv, logdet
    return inv


def svd_inverse(matrix, return_logdet=False):
    """SVD pseudo-inverse, singular values <= 1e-15 dropped (cosmogp/inv_matrix.py:4-18).

    HOST code on purpose: this is row a5 of SURVEY.md 8 -- a different algorithm the caller
    selects with svd_method=True, not part of the accelerated Cholesky path and never used as
    a substitute for it (8(f) rank 3 lists a device eigensolver as later work).  Like the
    reference it is SciPy's LAPACK dgesdd; the reference's progress print is dropped."""
    u, s, vt = linalg.svd(matrix)
    keep = s > 10 ** -15
    inv = np.dot(vt.T[:, keep], np.dot(np.diag(1. / s[keep]), u.T[keep]))
    if return_logdet:
        return inv, np.sum(np.log(s[keep]))
    return inv
