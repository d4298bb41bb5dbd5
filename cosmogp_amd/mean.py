"""Prior mean (mirror of cosmogp/mean.py) at the edge of the hot path: the result
is uploaded with y as the residual r = y - y0.  The 1D cubic spline is O(N) SciPy on the host like the reference; the
2D mean fits its bivariate spline on the host (bisplrep, O(knots)) and evaluates it at ALL points in one device launch
(cgp_bispev) where the reference calls bisplev once per point (mean.py:63-65)."""
import numpy as np
import scipy.interpolate as inter

#: which bisplrep task produced the last 2D fit: 1 = the reference's own call; 0 = the fallback below
LAST_BISPLREP_TASK = None


def interpolate_mean_1d(old_binning, mean_function, new_binning):
    """Cubic spline of the mean function on a new 1D grid (cosmogp/mean.py:7-33)."""
    return inter.InterpolatedUnivariateSpline(old_binning, mean_function)(new_binning)


def fit_mean_2d(old_binning, mean_function):
    """The reference's fit, cosmogp/mean.py:58: bisplrep(..., task=1).  task=1 continues a previous fit; called cold, as
    the reference does, it raises under recent SciPy (1.15 here: `Cannot produce output of size -4x-4`), so no output of
    the reference exists for this branch and its parity is UNPINNED.  When that call raises, the same smoothing spline
    is fitted from scratch (task=0, same defaults); LAST_BISPLREP_TASK says which one ran."""
    global LAST_BISPLREP_TASK
    old_binning = np.asarray(old_binning, dtype=np.float64)
    # (bisplrep ravels z: an (n, 1) column or any other layout of the n values works upstream, cosmogp/mean.py:58)
    mean_function = np.asarray(mean_function, dtype=np.float64).ravel()
    # genuine argument errors are reported as such, never papered over by the second attempt
    if old_binning.ndim != 2 or old_binning.shape[1] != 2 or mean_function.size != old_binning.shape[0]:
        raise ValueError('the 2D mean needs coordinates of shape (n, 2) and n mean values, got %r and %r'
                         % (old_binning.shape, mean_function.shape))
    if not (np.all(np.isfinite(old_binning)) and np.all(np.isfinite(mean_function))):
        raise ValueError('the 2D mean needs finite coordinates and values')
    try:
        tck = inter.bisplrep(old_binning[:, 0], old_binning[:, 1], mean_function, task=1)
        LAST_BISPLREP_TASK = 1
    except (RuntimeError, ValueError, TypeError) as cold:
        # only the cold-start failure of task=1 (no previous fit to continue) is answered by a fit from scratch; if that
        # fails as well the FIRST error is the one the caller sees
        try:
            tck = inter.bisplrep(old_binning[:, 0], old_binning[:, 1], mean_function, task=0)
        except (RuntimeError, ValueError, TypeError):
            raise cold
        LAST_BISPLREP_TASK = 0
    return tck


def interpolate_mean_2d(old_binning, mean_function, new_binning, device=None):
    """Bivariate spline of the mean function on new points (cosmogp/mean.py:36-67): host fit, device evaluation.  Like
    every product path of this package it needs the HIP library and a GPU: without them it raises _lib.HipUnavailable (there
    is deliberately no host evaluation to fall back to)."""
    from . import _lib
    tck = fit_mean_2d(old_binning, mean_function)
    dev = device if device is not None else _lib.default_device()
    return dev.bispev(tck, np.asarray(new_binning, dtype=np.float64))


def _is_1d(x):
    return type(x[0]) is np.float64          # the reference's dispatch, mean.py:81


def return_mean(y, x, new_x=None, mean_y=None, mean_xaxis=None, diff=None):
    """Mean on the data grid, or on new_x -- same contract as cosmogp/mean.py:70-104:
    y0 = mean_shape(x) + diff with diff = mean(y - mean_shape) unless given."""
    if mean_y is not None:
        assert mean_xaxis is not None, 'you should provide an x axis for the average'
        assert len(mean_y) == len(mean_xaxis), 'mean_y and mean_xaxis should have the same len'
        interp = interpolate_mean_1d if _is_1d(x) else interpolate_mean_2d
        shape = interp(mean_xaxis, mean_y, x)
    else:
        interp = None
        shape = 0
    if diff is None:
        diff = np.mean(y - shape)
    y0 = shape + diff
    if new_x is None:
        return y0
    if interp is None:
        return y0
    return interp(mean_xaxis, mean_y, new_x) + diff
