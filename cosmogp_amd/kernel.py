"""Host side of the covariance-assembly operators (mirror of cosmogp/kernel.py).

The O(N^2) work runs in the HIP kernels k_assemble_train / k_assemble_cross
(cosmogp_amd/csrc/kernels.hpp) through the C ABI entry point cgp_rbf.  What stays
on the host is the O(N) preparation, evaluated with the reference's own
expressions so the device sees bit-identical scalars:
  * sigma^2, l^2              cosmogp/kernel.py:72
  * inv_l (2D)                cosmogp/kernel.py:127-130
  * diagonal term             cosmogp/kernel.py:74-75 / :150-151
"""
import numpy as np

from . import _lib

#: The reference's 2D train-train branch omits sigma^2 (cosmogp/kernel.py:146) while its
#: cross branch has it (:141).  False reproduces the reference bit for bit; True gives the
#: mathematically consistent kernel.  Explicit switch, see DESIGN.md "Quirks".
SIGMA2_IN_TRAIN_2D = False


def init_rbf(x, y):
    """Crude first guess (sigma, L) -- mirror of cosmogp/kernel.py:6-22, host only.
    Like the reference, only the LAST object sets min/max/std (its loop overwrites
    scalars, kernel.py:14-17) while the point counts stay per object."""
    counts = np.zeros(len(y))
    lo = hi = sig = 0.
    for k in range(len(y)):
        lo, hi = np.min(x[k]), np.max(x[k])
        counts[k] = len(x[k])
        sig = np.std(y[k])
    spacing = np.mean(np.sqrt((hi - lo) ** 2 / counts))
    return np.mean(sig), np.mean([spacing, np.mean(hi - lo)])


def diag_term(n, y_err=None, floor=0., nugget=0.):
    """y_err**2 + floor**2 + nugget**2 in the reference's order (kernel.py:75)."""
    if y_err is None:
        y_err = np.zeros(n)
    return np.asarray(y_err, dtype=np.float64) ** 2 + floor ** 2 + nugget ** 2


def kernel_params_1d(hyperparameter, nugget=0.):
    """kp vector of include/cosmogp_hip.h for RBF1D; kp[6] = nugget**2 is added on the device to the
    resident diagonal vector y_err**2 + floor**2 (same left-to-right order as kernel.py:75)."""
    amp = hyperparameter[0] ** 2
    return np.array([amp, amp, hyperparameter[1] ** 2, 0., 0., 0., nugget ** 2], dtype=np.float64)


def kernel_params_2d(hyperparameter, sigma2_in_train=None, nugget=0.):
    """kp vector for RBF2D: inv_l built exactly as kernel.py:127-130; kp[6] = nugget**2."""
    if sigma2_in_train is None:
        sigma2_in_train = SIGMA2_IN_TRAIN_2D
    h = hyperparameter
    inv_l = np.array(([h[2] ** 2, -h[3]], [-h[3], h[1] ** 2]), dtype=np.float64)
    inv_l *= 1. / (((h[1] ** 2) * (h[2] ** 2)) - h[3] ** 2)
    amp = h[0] ** 2
    return np.array([amp if sigma2_in_train else 1.0, amp,
                     inv_l[0, 0], inv_l[0, 1], inv_l[1, 0], inv_l[1, 1], nugget ** 2], dtype=np.float64)


def _rbf(dim, x, kp, new_x, nugget, floor, y_err, device):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if len(x) == 0 or (new_x is not None and len(new_x) == 0):      # empty inputs: empty result, like NumPy
        return np.zeros((len(x) if new_x is None else len(new_x), len(x)))
    dev = device if device is not None else _lib.default_device()
    if new_x is None:
        return dev.rbf(dim, x, kp, diag=diag_term(len(x), y_err, floor, nugget))
    return dev.rbf(dim, x, kp, xnew=np.ascontiguousarray(new_x, dtype=np.float64))


def rbf_kernel_1d(x, hyperparameter, new_x=None, nugget=0., floor=0.00, y_err=None, device=None):
    """1D RBF covariance on the GPU; same signature and result layout as
    cosmogp/kernel.py:25-77: (N, N) with the diagonal term when new_x is None,
    else the (M, N) cross matrix (rows = new points, no diagonal term)."""
    return _rbf(1, x, kernel_params_1d(hyperparameter), new_x, nugget, floor, y_err, device)


def rbf_kernel_2d(x, hyperparameter, new_x=None, nugget=0., floor=0.00, y_err=None, device=None):
    """Anisotropic 2D RBF covariance on the GPU; same signature as cosmogp/kernel.py:80-155
    (hyperparameter = [sigma, lx, ly, lxy]).  The reference's debug prints are not reproduced;
    its missing sigma^2 in the train-train branch is (see SIGMA2_IN_TRAIN_2D)."""
    return _rbf(2, x, kernel_params_2d(hyperparameter), new_x, nugget, floor, y_err, device)
