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


def gradient_from_sums_1d(hyperparameter, nugget, sums):
    """Chain rule from the moment sums of cgp_fit_grad to d logL / d [sigma, l, nugget] for RBF1D
    (K = sigma^2 exp(-d^2 / (2 l^2)) + diag, kernel.py:72-75)."""
    sig, ell = hyperparameter[0], hyperparameter[1]
    return np.array([sig * sums[0], 0.5 * sig ** 2 * sums[1] / ell ** 3, nugget * sums[4]])


def gradient_from_sums_2d(hyperparameter, nugget, sums, sigma2_in_train=None):
    """Same for RBF2D, d logL / d [sigma, lx, ly, lxy, nugget].  inv_l = L^-1 with
    L = [[lx^2, lxy], [lxy, ly^2]] (kernel.py:127-130), so d inv_l = -inv_l dL inv_l.  With the
    reference's train-train matrix (no sigma^2, kernel.py:146) the likelihood does not depend on sigma."""
    if sigma2_in_train is None:
        sigma2_in_train = SIGMA2_IN_TRAIN_2D
    h = hyperparameter
    kp = kernel_params_2d(h, sigma2_in_train=sigma2_in_train)
    vi = np.array([[kp[2], kp[3]], [kp[4], kp[5]]])
    g_vi = -0.25 * kp[0] * np.array([[sums[1], sums[2]], [sums[2], sums[3]]])        # d logL / d inv_l
    out = np.zeros(5)
    out[0] = h[0] * sums[0] if sigma2_in_train else 0.
    for k, dl in ((1, np.array([[2. * h[1], 0.], [0., 0.]])), (2, np.array([[0., 0.], [0., 2. * h[2]]])),
                  (3, np.array([[0., 1.], [1., 0.]]))):
        out[k] = np.sum(g_vi * (-vi.dot(dl).dot(vi)))
    out[4] = nugget * sums[4]
    return out


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
