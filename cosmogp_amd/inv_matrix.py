"""Mirror of cosmogp/inv_matrix.py: the two callables of the reference's `chol` / `svd` seam
(cosmogp/Gaussian_process.py:6-9)."""
import numpy as np

from . import _lib

#: singular values at or below this are dropped (cosmogp/inv_matrix.py:7)
SVD_THRESHOLD = 10 ** -15


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


def svd_inverse(matrix, return_logdet=False, device=None):
    """Pseudo-inverse with the singular values <= 1e-15 dropped, on the GPU -- same contract as
    cosmogp/inv_matrix.py:4-18 (never raises on singular input; logdet = sum of log of the kept
    singular values; the reference's progress print is dropped).

    The matrices of this path are symmetric, so the SVD is the eigen-decomposition
    (s = |lambda|, U = V sign(lambda)): the device runs a two-sided block Jacobi eigensolver
    (csrc/jacobi.hpp) and forms V diag(1/lambda) V' with the MFMA GEMM.  For positive definite K the
    result equals the reference's dgesdd route to ~cond(K) eps; for numerically singular K the modes at
    the rounding-noise level differ between any two SVD implementations, and so does which of them pass
    the 1e-15 test."""
    dev = device if device is not None else _lib.default_device()
    inv, logdet, _ = dev.svd_inverse(np.asarray(matrix, dtype=np.float64), SVD_THRESHOLD)
    if return_logdet:
        return inv, logdet
    return inv
