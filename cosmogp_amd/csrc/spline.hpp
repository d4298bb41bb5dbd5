// Tensor-product B-spline evaluation at scattered points: the device form of the loop
// cosmogp/mean.py:63-65 (`bisplev` once per point).  One lane per point, FITPACK's own algorithm (fpbisp / fpbspl):
// clamp the argument to the base interval, locate the knot span, de Boor's stable recurrence for the k + 1 non-zero
// basis functions in each direction, then the (kx + 1) x (ky + 1) coefficient window.  The knot vectors and the
// coefficient array come from the host fit (scipy.interpolate.bisplrep): O(number of knots), not worth a kernel.
#pragma once
#include <hip/hip_runtime.h>

#define CGP_SPL_KMAX 5

// h[0..k] = the k + 1 non-zero B-splines of degree k at x, knot span t[l] <= x < t[l+1] (0-based l)
__device__ __forceinline__ void spl_basis(const double* __restrict__ t, int k, double x, int l, double* h) {
    double hh[CGP_SPL_KMAX + 1];
    h[0] = 1.;
    for (int j = 1; j <= k; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.;
        for (int i = 0; i < j; ++i) {
            const int li = l + i + 1, lj = li - j;
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] += f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

__device__ __forceinline__ int spl_span(const double* __restrict__ t, int n, int k, double& x) {
    const double tb = t[k], te = t[n - k - 1];
    x = x < tb ? tb : (x > te ? te : x);            // fpbisp: arguments outside the base interval take the boundary value
    int l = k;
    while (l < n - k - 2 && x >= t[l + 1]) ++l;     // (a handful of knots: linear search like fpbisp)
    return l;
}

// out[p] = sum_ij c[(lx - kx + i) * (ny - ky - 1) + (ly - ky + j)] Bx_i(x_p) By_j(y_p);  pts = m x 2 row-major
__global__ __launch_bounds__(256) void k_bispev(const double* __restrict__ tx, int nx, const double* __restrict__ ty, int ny,
                                                const double* __restrict__ c, int kx, int ky, const double* __restrict__ pts,
                                                long m, double* __restrict__ out) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    double x = pts[2 * p], y = pts[2 * p + 1];
    const int lx = spl_span(tx, nx, kx, x), ly = spl_span(ty, ny, ky, y);
    double hx[CGP_SPL_KMAX + 1], hy[CGP_SPL_KMAX + 1];
    spl_basis(tx, kx, x, lx, hx);
    spl_basis(ty, ky, y, ly, hy);
    const int nky1 = ny - ky - 1;
    double s = 0.;
    for (int i = 0; i <= kx; ++i) {
        double si = 0.;
        for (int j = 0; j <= ky; ++j) si += c[(long)(lx - kx + i) * nky1 + (ly - ky + j)] * hy[j];
        s += si * hx[i];
    }
    out[p] = s;
}
