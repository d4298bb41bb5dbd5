"""cosmogp_amd -- MI355X-native (gfx950) implementation of the cosmogp GP hot path.

Same import surface as the reference package (cosmogp/__init__.py:12-25) so that
`import cosmogp_amd as cosmogp` is a drop-in for RBF covariance assembly, the Cholesky
inverse / solve and the Gaussian_process fit / predict / log-likelihood that sit on them.
All O(N^2) / O(N^3) work runs in hand-written HIP kernels behind the C ABI of
include/cosmogp_hip.h (cosmogp_amd/libcosmogp_hip.so, loaded with ctypes).  There is no CPU
fallback for that path: without the library or a GPU the calls raise HipUnavailable.
"""
from .inv_matrix import svd_inverse
from .inv_matrix import cholesky_inverse

from .mean import return_mean

from .gaussian_process import Gaussian_process
from .gaussian_process import gaussian_process
from .gaussian_process import gaussian_process_nobject
from .gaussian_process import gaussian_process_2d
from .gaussian_process import log_likelihood_gp

from .kernel import init_rbf
from .kernel import rbf_kernel_1d
from .kernel import rbf_kernel_2d

from .pull import build_pull

from ._lib import Device, HipUnavailable, HipError, default_device

__all__ = ['svd_inverse', 'cholesky_inverse', 'return_mean', 'Gaussian_process', 'gaussian_process',
           'gaussian_process_nobject', 'gaussian_process_2d', 'log_likelihood_gp', 'init_rbf',
           'rbf_kernel_1d', 'rbf_kernel_2d', 'build_pull', 'Device', 'HipUnavailable', 'HipError',
           'default_device']
