"""Prior mean (mirror of cosmogp/mean.py) -- O(N) host splines at the edge of the hot path
(SURVEY.md 8a row a13): the result is uploaded with y as the residual r = y - y0."""
import numpy as np
import scipy.interpolate as inter


def interpolate_mean_1d(old_binning, mean_function, new_binning):
    """Cubic spline of the mean function on a new 1D grid (cosmogp/mean.py:7-33)."""
    return inter.InterpolatedUnivariateSpline(old_binning, mean_function)(new_binning)


def interpolate_mean_2d(old_binning, mean_function, new_binning):
    """Bivariate spline (bisplrep task=1) evaluated point by point (cosmogp/mean.py:36-67)."""
    tck = inter.bisplrep(old_binning[:, 0], old_binning[:, 1], mean_function, task=1)
    out = np.zeros(len(new_binning))
    for k in range(len(new_binning)):
        out[k] = inter.bisplev(new_binning[k, 0], new_binning[k, 1], tck)
    return out


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
