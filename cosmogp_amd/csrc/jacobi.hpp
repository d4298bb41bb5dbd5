// Device symmetric eigensolver for the reference's svd_method=True route (cosmogp/inv_matrix.py:4-18:
// SVD pseudo-inverse with singular values <= 1e-15 dropped).  For the symmetric K of this path the SVD
// is the eigen-decomposition (s = |lambda|, U = V sign(lambda)), so
//     pinv = sum_{|lambda_p| > thr} v_p v_p' / lambda_p ,   logdet = sum log |lambda_p| .
//
// Algorithm: two-sided BLOCK JACOBI.  K (Np x Np, both triangles, identity-padded) is cut into
// nb = Np / 32 block rows/columns; a round of the round-robin tournament pairs all blocks (nb / 2
// disjoint pairs), and for every pair (I, J)
//   E1  k_jac_inner : the 64 x 64 pivot matrix S = K[IJ, IJ] is diagonalised completely in LDS by
//                     cyclic Jacobi with the parallel ordering (63 steps x 32 disjoint rotations per
//                     sweep), Q accumulated in LDS                       -> Q_pair (64 x 64)
//   E2  k_jac_apply<0>: K[IJ, :] <- Q' K[IJ, :]      (all pairs: disjoint rows)
//   E3  k_jac_apply<1>: K[:, IJ] <- K[:, IJ] Q  and  V[:, IJ] <- V[:, IJ] Q   (disjoint columns)
// nb - 1 rounds = one sweep; sweeps repeat until no rotation larger than tol = eps ||K||_F was needed.
// E2 / E3 are thin (depth 64) updates, HBM/L2-bound: fp64 VALU FMAs from LDS-staged 64 x 64 tiles (the
// vector and matrix fp64 peaks of this part are equal; MFMA would buy nothing at depth 64).
// Rotations with |s_pq| <= tol are skipped, so eigenvalues carry an absolute error of a few eps ||K||,
// like LAPACK's; modes at that noise level (numerically singular K) come out smaller than dgesdd's
// and are dropped by the same 1e-15 test the reference applies (documented in DESIGN.md).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

#define JB 32            // block size
#define JP 64            // pivot matrix size (two blocks)
#define JLD 66           // LDS row stride (doubles): 16-B aligned rows, column walks spread over banks

struct JacStat {
    double fro;          // ||K||_F at the start
    double tol;          // rotation threshold (absolute)
    int rotations;       // pivot matrices that needed at least one rotation in the current sweep
    int pad;
};

// ||K||_F and the threshold in two launches: per-workgroup partial sums of squares in a fixed order (the result does not
// depend on the grid's scheduling), then one workgroup adds the partials
#define JAC_NORM_BLOCKS 1024
// (only the n x n matrix proper: the identity padding up to Np must not enter the norm -- for a matrix of small scale, e.g.
// sigma^2 = 2e-5, the padding's ones would set the threshold, and rotations the size of the matrix's own small eigenvalues
// would be skipped: the round-3 noise-free golden came out with one kept mode missing until this was fixed)
__global__ __launch_bounds__(256) void k_jac_norm_partial(const double* __restrict__ K, long ld, int Np, int n, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.;
    const long half = Np / 2, tot = (long)n * half;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long)JAC_NORM_BLOCKS * 256) {
        const long r = idx / half, c = (idx - r * half) * 2;
        const double2 v = *(const double2*)(K + r * ld + c);
        if (c < n) s = fma(v.x, v.x, s);
        if (c + 1 < n) s = fma(v.y, v.y, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void k_jac_norm(const double* __restrict__ part, double tol_factor, JacStat* __restrict__ st) {
    __shared__ double red[16];
    double s = part[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.;
        for (int i = 0; i < 16; ++i) t += red[i];
        st->fro = sqrt(t);
        st->tol = tol_factor * 2.220446049250313e-16 * sqrt(t);
        st->rotations = 0;
    }
}

__global__ void k_jac_identity(double* __restrict__ V, long ld, int Np) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)Np * Np;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / Np, c = idx - r * Np;
        V[r * ld + c] = (r == c) ? 1. : 0.;
    }
}

__device__ __forceinline__ int jac_row(int I, int J, int r) { return r < JB ? I * JB + r : J * JB + (r - JB); }

// E1: one workgroup per pair.  pairs[2 k], pairs[2 k + 1] = block indices.  Q_pair -> Qws[k][64][64]
// (row-major, Q[s][c]); flags[k] = 1 when any rotation was applied (0: Q = I, the updates skip the pair).
__global__ __launch_bounds__(256) void k_jac_inner(const double* __restrict__ K, long ld, const int* __restrict__ pairs,
                                                   double* __restrict__ Qws, int* __restrict__ flags,
                                                   JacStat* __restrict__ st, int max_sweeps) {
    __shared__ __attribute__((aligned(16))) double S[JP * JLD];
    __shared__ __attribute__((aligned(16))) double Q[JP * JLD];
    __shared__ double rc[32], rs[32];
    __shared__ int rp[32], rq[32];
    __shared__ int any_sweep, any_total;
    const int k = blockIdx.x, tid = threadIdx.x;
    const int I = pairs[2 * k], J = pairs[2 * k + 1];
    const double tol = st->tol;
    for (int idx = tid; idx < JP * JP; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        S[r * JLD + c] = K[(long)jac_row(I, J, r) * ld + jac_row(I, J, c)];
        Q[r * JLD + c] = (r == c) ? 1. : 0.;
    }
    if (tid == 0) any_total = 0;
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) any_sweep = 0;
        __syncthreads();
        for (int step = 0; step < JP - 1; ++step) {
            if (tid < 32) {
                // circle method on 64 players: player 63 stays, the others rotate
                int p, q;
                if (tid == 0) { p = step; q = JP - 1; }
                else { p = (step + tid) % (JP - 1); q = (step - tid + (JP - 1)) % (JP - 1); }
                if (p > q) { int t = p; p = q; q = t; }
                const double apq = S[p * JLD + q];
                double c = 1., s = 0.;
                if (fabs(apq) > tol) {
                    const double tau = (S[q * JLD + q] - S[p * JLD + p]) / (2. * apq);
                    const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
                    c = 1. / sqrt(1. + t * t);
                    s = t * c;
                    any_sweep = 1;
                }
                rc[tid] = c; rs[tid] = s; rp[tid] = p; rq[tid] = q;
            }
            __syncthreads();
            // rows: S <- J' S
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int item = tid + 256 * i, pr = item >> 6, col = item & 63;
                const double s = rs[pr];
                if (s != 0.) {
                    const double c = rc[pr];
                    const int p = rp[pr], q = rq[pr];
                    const double a = S[p * JLD + col], b = S[q * JLD + col];
                    S[p * JLD + col] = c * a - s * b;
                    S[q * JLD + col] = s * a + c * b;
                }
            }
            __syncthreads();
            // columns: S <- S J,  Q <- Q J
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int item = tid + 256 * i, pr = item >> 6, row = item & 63;
                const double s = rs[pr];
                if (s != 0.) {
                    const double c = rc[pr];
                    const int p = rp[pr], q = rq[pr];
                    double a = S[row * JLD + p], b = S[row * JLD + q];
                    S[row * JLD + p] = c * a - s * b;
                    S[row * JLD + q] = s * a + c * b;
                    a = Q[row * JLD + p]; b = Q[row * JLD + q];
                    Q[row * JLD + p] = c * a - s * b;
                    Q[row * JLD + q] = s * a + c * b;
                }
            }
            __syncthreads();
        }
        const int more = any_sweep;
        __syncthreads();
        if (!more) break;
        if (tid == 0) any_total = 1;
    }
    __syncthreads();
    const int rotated = any_total;
    if (rotated) {
        double* Qo = Qws + (long)k * JP * JP;
        for (int idx = tid; idx < JP * JP; idx += 256) Qo[idx] = Q[(idx >> 6) * JLD + (idx & 63)];
    }
    if (tid == 0) {
        flags[k] = rotated;
        if (rotated) atomicAdd(&st->rotations, 1);
    }
}

// Small matrices (n <= 64): the whole problem is ONE pivot matrix.  A single workgroup diagonalises the leading
// 64 x 64 block of K (identity-padded beyond n) in LDS until no rotation above the threshold is left, then writes
// the eigenvalues onto the diagonal of K (off-diagonals of that block := 0) and V[0:64, 0:64] := Q.  Everything
// outside the block is the identity padding already (V was initialised to I), so k_jac_finish and the K^+ GEMM run
// unchanged.  Replaces (Np / 32 - 1) rounds x ~15 sweeps x 3 launches by one launch.
__global__ __launch_bounds__(256) void k_jac_small(double* __restrict__ K, double* __restrict__ V, long ld,
                                                   const JacStat* __restrict__ st, int max_sweeps) {
    __shared__ __attribute__((aligned(16))) double S[JP * JLD];
    __shared__ __attribute__((aligned(16))) double Q[JP * JLD];
    __shared__ double rc[32], rs[32];
    __shared__ int rp[32], rq[32];
    __shared__ int any_sweep;
    const int tid = threadIdx.x;
    const double tol = st->tol;
    for (int idx = tid; idx < JP * JP; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        S[r * JLD + c] = K[(long)r * ld + c];
        Q[r * JLD + c] = (r == c) ? 1. : 0.;
    }
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) any_sweep = 0;
        __syncthreads();
        for (int step = 0; step < JP - 1; ++step) {
            if (tid < 32) {
                int p, q;
                if (tid == 0) { p = step; q = JP - 1; }
                else { p = (step + tid) % (JP - 1); q = (step - tid + (JP - 1)) % (JP - 1); }
                if (p > q) { int t = p; p = q; q = t; }
                const double apq = S[p * JLD + q];
                double c = 1., s = 0.;
                if (fabs(apq) > tol) {
                    const double tau = (S[q * JLD + q] - S[p * JLD + p]) / (2. * apq);
                    const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
                    c = 1. / sqrt(1. + t * t);
                    s = t * c;
                    any_sweep = 1;
                }
                rc[tid] = c; rs[tid] = s; rp[tid] = p; rq[tid] = q;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int item = tid + 256 * i, pr = item >> 6, col = item & 63;
                const double s = rs[pr];
                if (s != 0.) {
                    const double c = rc[pr];
                    const int p = rp[pr], q = rq[pr];
                    const double a = S[p * JLD + col], b = S[q * JLD + col];
                    S[p * JLD + col] = c * a - s * b;
                    S[q * JLD + col] = s * a + c * b;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int item = tid + 256 * i, pr = item >> 6, row = item & 63;
                const double s = rs[pr];
                if (s != 0.) {
                    const double c = rc[pr];
                    const int p = rp[pr], q = rq[pr];
                    double a = S[row * JLD + p], b = S[row * JLD + q];
                    S[row * JLD + p] = c * a - s * b;
                    S[row * JLD + q] = s * a + c * b;
                    a = Q[row * JLD + p]; b = Q[row * JLD + q];
                    Q[row * JLD + p] = c * a - s * b;
                    Q[row * JLD + q] = s * a + c * b;
                }
            }
            __syncthreads();
        }
        const int more = any_sweep;
        __syncthreads();
        if (!more) break;
    }
    for (int idx = tid; idx < JP * JP; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        K[(long)r * ld + c] = (r == c) ? S[r * JLD + r] : 0.;
        V[(long)r * ld + c] = Q[r * JLD + c];
    }
}

// E2 (MODE 0): X[IJ, c0:c0+64] <- Q' X[IJ, c0:c0+64]        grid (Np / 64, npairs, 1),     X = K
// E3 (MODE 1): X[r0:r0+64, IJ] <- X[r0:r0+64, IJ] Q           grid (Np / 64, npairs, 2),     X = K, V
// Both are  Y[a][b] = sum_s A[s][a] B[s][b]  on LDS tiles:  MODE 0: A = Q, B = X;  MODE 1: A = X^T, B = Q.
// Thread (ta, tb) owns the 4 x 4 outputs a = 4 ta.., b = 4 tb...
template <int MODE>
__global__ __launch_bounds__(256) void k_jac_apply(double* __restrict__ X0, double* __restrict__ X1, long ld,
                                                   const int* __restrict__ pairs, const double* __restrict__ Qws,
                                                   const int* __restrict__ flags) {
    const int k = blockIdx.y;
    if (!flags[k]) return;
    __shared__ __attribute__((aligned(16))) double As[JP * JLD];
    __shared__ __attribute__((aligned(16))) double Bs[JP * JLD];
    double* X = blockIdx.z ? X1 : X0;
    const int I = pairs[2 * k], J = pairs[2 * k + 1];
    const int tid = threadIdx.x, o0 = blockIdx.x * 64;
    const double* Qg = Qws + (long)k * JP * JP;
    // Q[s][c] -> (MODE 0) As[s][c]  /  (MODE 1) Bs[s][c]
    double* Qdst = MODE == 0 ? As : Bs;
    for (int idx = tid; idx < JP * JP / 2; idx += 256) {
        const int s = idx >> 5, c = (idx & 31) * 2;
        *(double2*)(Qdst + s * JLD + c) = *(const double2*)(Qg + s * JP + c);
    }
    if (MODE == 0) {
        // Bs[s][b] = X[row(s)][o0 + b]
        for (int idx = tid; idx < JP * JP / 2; idx += 256) {
            const int s = idx >> 5, b = (idx & 31) * 2;
            *(double2*)(Bs + s * JLD + b) = *(const double2*)(X + (long)jac_row(I, J, s) * ld + o0 + b);
        }
    } else {
        // As[s][a] = X[o0 + a][col(s)]   (transposed while staging)
        for (int idx = tid; idx < JP * JP / 2; idx += 256) {
            const int a = idx >> 5, s = (idx & 31) * 2;
            const double2 v = *(const double2*)(X + (long)(o0 + a) * ld + jac_row(I, J, s));
            As[s * JLD + a] = v.x;
            As[(s + 1) * JLD + a] = v.y;
        }
    }
    __syncthreads();
    const int ta = tid >> 4, tb = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.;
#pragma unroll 8
    for (int s = 0; s < JP; ++s) {
        const double2 a0 = *(const double2*)(As + s * JLD + 4 * ta), a1 = *(const double2*)(As + s * JLD + 4 * ta + 2);
        const double2 b0 = *(const double2*)(Bs + s * JLD + 4 * tb), b1 = *(const double2*)(Bs + s * JLD + 4 * tb + 2);
        const double a[4] = {a0.x, a0.y, a1.x, a1.y}, b[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = X + (long)jac_row(I, J, 4 * ta + i) * ld + o0 + 4 * tb;
            *(double2*)dst = make_double2(acc[i][0], acc[i][1]);
            *(double2*)(dst + 2) = make_double2(acc[i][2], acc[i][3]);
        }
    } else {
        // 4 tb .. 4 tb + 3 never straddles the two column blocks (32 is a multiple of 4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = X + (long)(o0 + 4 * ta + i) * ld + jac_row(I, J, 4 * tb);
            *(double2*)dst = make_double2(acc[i][0], acc[i][1]);
            *(double2*)(dst + 2) = make_double2(acc[i][2], acc[i][3]);
        }
    }
}

// lambda_p = K[p][p]; w_p = 1 / lambda_p when |lambda_p| > thresh else 0; VD[:, p] = V[:, p] w_p.
// res[0] = logdet = sum over kept of log |lambda_p|, res[1] = number of dropped modes among p < n
// (padding modes are exact 1s: kept, log 1 = 0).  One workgroup computes the scalars (block 0), all scale.
// power = 2 (low-rank route below): w_p = 1 / lambda_p^2; extra = modes already known to be dropped (added to res[1]).
__global__ __launch_bounds__(256) void k_jac_finish(const double* __restrict__ K, const double* __restrict__ V,
                                                    double* __restrict__ VD, long ld, int Np, int n, double thresh,
                                                    double* __restrict__ lam_out, double* __restrict__ res, int power = 1,
                                                    double extra = 0.) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ double red[2][4];
        double lg = 0., nd = 0.;
        for (int p = tid; p < Np; p += 256) {
            const double l = K[(long)p * ld + p];
            lam_out[p] = l;
            if (fabs(l) > thresh) lg += log(fabs(l));
            else if (p < n) nd += 1.;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lg += __shfl_down(lg, off); nd += __shfl_down(nd, off); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = lg; red[1][tid >> 6] = nd; }
        __syncthreads();
        if (tid == 0) {
            res[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            res[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3] + extra;
        }
    }
    const long tot = (long)Np * Np / 2;
    for (long idx = (long)blockIdx.x * 256 + tid; idx < tot; idx += (long)gridDim.x * 256) {
        const long r = idx / (Np / 2), c = (idx - r * (Np / 2)) * 2;
        const double l0 = K[c * ld + c], l1 = K[(c + 1) * ld + c + 1];
        const double2 v = *(const double2*)(V + r * ld + c);
        const double w0 = fabs(l0) > thresh ? (power == 2 ? 1. / (l0 * l0) : 1. / l0) : 0.;
        const double w1 = fabs(l1) > thresh ? (power == 2 ? 1. / (l1 * l1) : 1. / l1) : 0.;
        *(double2*)(VD + r * ld + c) = make_double2(power == 2 ? v.x * w0 : (fabs(l0) > thresh ? v.x / l0 : 0.),
                                                    power == 2 ? v.y * w1 : (fabs(l1) > thresh ? v.y / l1 : 0.));
    }
}

// upper triangle := mirror of the lower one (in place)
__global__ void k_mirror_lower(double* __restrict__ M, long ld, int Np) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)Np * Np;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / Np, c = idx - r * Np;
        if (c > r) M[r * ld + c] = M[c * ld + r];
    }
}

// Finish of the pseudo-inverse route.  tall = [H (m rows); zeros; r (row zrow)], T = tall K^+ (same shape):
// block t < m:  mu_t = H_t . alpha + mean_t  (alpha = row zrow of T),  var_t = (amp_train + kss_noise) - T_t . H_t
// block m:      loglik = -1/2 r . alpha - n/2 log(2 pi) - 1/2 logdet   (Gaussian_process.py:68-73)
__global__ __launch_bounds__(256) void k_finish_pinv(
    const double* __restrict__ tall, const double* __restrict__ T, long ld, int Np, int zrow, int m,
    const double* __restrict__ mean_new, const KParams* __restrict__ kpp, const double* __restrict__ res,
    const int* __restrict__ ns, double* __restrict__ mu, double* __restrict__ var, double* __restrict__ scal) {
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* alpha = T + (long)zrow * ld;
    __shared__ double red[2][4];
    double s1 = 0., s2 = 0.;
    if (t < m) {
        const double* hrow = tall + (long)t * ld;
        const double* trow = T + (long)t * ld;
        for (int c = tid * 2; c < Np; c += 512) {
            const double2 hv = *(const double2*)(hrow + c), av = *(const double2*)(alpha + c), tv = *(const double2*)(trow + c);
            s1 += hv.x * av.x + hv.y * av.y;
            s2 += hv.x * tv.x + hv.y * tv.y;
        }
    } else {
        const double* r = tall + (long)zrow * ld;
        for (int c = tid * 2; c < Np; c += 512) {
            const double2 rv = *(const double2*)(r + c), av = *(const double2*)(alpha + c);
            s1 += rv.x * av.x + rv.y * av.y;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (t < m) {
            mu[t] = s1 + (mean_new ? mean_new[t] : 0.);
            var[t] = (kpp->amp_train + kpp->kss_noise) - s2;
        } else {
            const double n = (double)ns[0];
            scal[0] = -0.5 * s1 - 0.5 * n * 1.8378770664093453 - 0.5 * res[0];
            scal[1] = res[0];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Low-rank pre-stage of the eigen route (round 3).  The route runs when the data carry no noise (with noise the Cholesky route
// serves svd_method=True, gaussian_process.SVD_VIA_CHOLESKY_WHEN_PD), and a noise-free RBF matrix is numerically of LOW RANK:
// N = 4096 points on a domain of 200 length scales have 541 eigenvalues above 1e-13 |K| -- everything below is the rounding
// noise of any decomposition.  Diagonally pivoted Cholesky  K = G G^T + E,  max_i E_ii <= tol = 4 eps max_i K_ii, after r << N
// steps; then the r x r problem  G^T G = U S U^T  (same non-zero spectrum as G G^T) on the Jacobi sweeps above, and
//     K^+ = G U S^-2 U^T G^T   (modes S_p > thresh),   logdet = sum log S_p,   dropped = (n - r) + #{S_p <= thresh}.
// One kernel per step: every workgroup re-reduces the previous step's per-workgroup maxima of the residual diagonal d (pivot
// p), computes its 64 rows of column j,  g_i = (K_ip - sum_{c<j} G_ic G_pc) / sqrt(d_p)  (G stored transposed: row c of Gt is
// column c of G, contiguous in i), updates d and leaves its own maximum for the next step.  A matrix that does not reach
// the tolerance within rcap steps (full rank: noisy data with the eigen route forced) goes to the full sweeps unchanged.
// ------------------------------------------------------------------------------------------------------------------
struct PcholCtl {
    int done;            // 1: the residual diagonal fell below the tolerance at step `rank`
    int rank;
    double tol;          // absolute tolerance on the residual diagonal
    double dmax0;        // largest diagonal entry at the start
    double dcur;         // largest residual diagonal entry at the last step taken (the host gives up on matrices of full rank)
};

#define PCHOL_ROWS 64

// d := diag(K) (rows < n; -1 beyond: never a pivot) and the maximum of every 64-row group; ONE workgroup (the diagonal is N
// doubles), which also sets the tolerance 4 eps max_i K_ii
__global__ __launch_bounds__(256) void k_pchol_init(const double* __restrict__ K, long ld, int n, int Np, double* __restrict__ d,
                                                    double2* __restrict__ part, PcholCtl* __restrict__ ctl, int nparts) {
    __shared__ double sv[256];
    __shared__ int si[256];
    const int tid = threadIdx.x;
    double best = -1.;
    for (int g0 = 0; g0 < nparts; g0 += 4) {
        // four groups of 64 rows per pass: thread tid looks at row (g0 + tid / 64) * 64 + tid % 64
        const int i = g0 * PCHOL_ROWS + tid;
        const double v = i < n ? K[(long)i * ld + i] : -1.;
        if (i < Np) d[i] = v;
        sv[tid] = v; si[tid] = i;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {            // reduce inside each group of 64 (lowest index wins ties)
            if ((tid & 63) < off && (sv[tid + off] > sv[tid] || (sv[tid + off] == sv[tid] && si[tid + off] < si[tid]))) {
                sv[tid] = sv[tid + off]; si[tid] = si[tid + off];
            }
            __syncthreads();
        }
        if ((tid & 63) == 0 && g0 + (tid >> 6) < nparts) part[g0 + (tid >> 6)] = make_double2(sv[tid], (double)si[tid]);
        if (tid == 0)
            for (int q = 0; q < 4; ++q) best = fmax(best, sv[64 * q]);
        __syncthreads();
    }
    if (tid == 0) {
        ctl->done = 0; ctl->rank = 0;
        ctl->dmax0 = best;
        ctl->dcur = best;
        ctl->tol = 4. * 2.220446049250313e-16 * best;
    }
}

// step j: grid = Np / PCHOL_ROWS workgroups of 256 threads (64 rows x 4 partitions of the sum over c)
__global__ __launch_bounds__(256) void k_pchol_step(const double* __restrict__ K, long ld, int n, double* __restrict__ Gt, long ldg,
                                                    double* __restrict__ d, const double2* __restrict__ part_in,
                                                    double2* __restrict__ part_out, int nparts, int j, PcholCtl* __restrict__ ctl) {
    __shared__ double sacc[4][PCHOL_ROWS];
    __shared__ double sv[PCHOL_ROWS];
    __shared__ int si[PCHOL_ROWS];
    __shared__ double s_p[2];
    __shared__ int s_done;
    const int tid = threadIdx.x, rr = tid & 63, q = tid >> 6;
    // `done` may be set by workgroup 0 of THIS launch while later workgroups start: ONE thread reads it and the whole
    // workgroup follows that value (every thread reading for itself could split a workgroup around the barrier below)
    if (tid == 0) { s_done = ctl->done; s_p[0] = -1.; s_p[1] = 0.; }
    __syncthreads();
    if (s_done) return;
    // pivot: the largest residual diagonal entry (lowest index on ties), the same in every workgroup
    if (tid < 64) {
        double m = -1.;
        int mi = 0x7fffffff;
        for (int g = tid; g < nparts; g += 64) {
            const double2 v = part_in[g];
            if (v.x > m || (v.x == m && (int)v.y < mi)) { m = v.x; mi = (int)v.y; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double om = __shfl_down(m, off);
            const int oi = __shfl_down(mi, off);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        if (tid == 0) { s_p[0] = m; s_p[1] = (double)mi; }
    }
    __syncthreads();
    const double dp = s_p[0];
    const int p = (int)s_p[1];
    if (!(dp > ctl->tol)) {
        if (blockIdx.x == 0 && tid == 0) { ctl->rank = j; ctl->done = 1; }
        return;
    }
    if (blockIdx.x == 0 && tid == 0) ctl->dcur = dp;
    const int i = blockIdx.x * PCHOL_ROWS + rr;
    double acc = 0.;
    for (int c = q; c < j; c += 4) acc = fma(Gt[(long)c * ldg + i], Gt[(long)c * ldg + p], acc);
    sacc[q][rr] = acc;
    __syncthreads();
    if (tid < PCHOL_ROWS) {
        const double s = (sacc[0][rr] + sacc[1][rr]) + (sacc[2][rr] + sacc[3][rr]);
        double g = 0., dn = -1.;
        if (i < n) {
            g = (K[(long)p * ld + i] - s) / sqrt(dp);           // (K symmetric: row p is column p, contiguous in i)
            const double dold = d[i];
            if (dold < 0.) dn = dold;                              // an earlier pivot stays out
            else if (i == p) dn = -1.;                             // the pivot is never chosen again
            else dn = fmax(dold - g * g, 0.);                      // (rounding below zero: exhausted, not negative)
            d[i] = dn;
        }
        Gt[(long)j * ldg + i] = g;
        sv[rr] = dn; si[rr] = i;
    }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        if (tid < off && (sv[tid + off] > sv[tid] || (sv[tid + off] == sv[tid] && si[tid + off] < si[tid]))) {
            sv[tid] = sv[tid + off]; si[tid] = si[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) part_out[blockIdx.x] = make_double2(sv[0], (double)si[0]);
}

// dst (cols x rows, ld = ldd) := src^T (rows x cols, ld = lds), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ src, long lds_, double* __restrict__ dst, long ldd, int rows, int cols) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (by + r < rows && bx + tx < cols) t[r][tx] = src[(long)(by + r) * lds_ + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) dst[(long)(bx + r) * ldd + by + tx] = t[tx][r];
}

// A (rp x rp, lower triangle given) -> full symmetric with the identity on the padding indices >= r
__global__ void k_sym_pad(double* __restrict__ A, long ld, int r, int rp) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)rp * rp;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        const long i = idx / rp, c = idx - i * rp;
        if (i >= r || c >= r) A[i * ld + c] = i == c ? 1. : 0.;
        else if (c > i) A[i * ld + c] = A[c * ld + i];
    }
}

// rows / columns >= n of an Np x Np matrix := those of the identity (the padding of the pseudo-inverse)
__global__ void k_pad_identity(double* __restrict__ A, long ld, int n, int Np) {
    const long npad = Np - n;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = npad * Np;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        const long q = idx / Np, c = idx - q * Np, i = n + q;
        A[i * ld + c] = i == c ? 1. : 0.;           // row i >= n
        if (c < n) A[c * ld + i] = 0.;              // column i >= n above the padding rows
    }
}

// Full-rank shortcut of the eigen route (api.hip, chol_certified_pinv).  For a symmetric positive definite K,
// lambda_min(K) = 1 / ||K^-1||_2 >= 1 / ||K^-1||_F: when that bound clears the threshold no singular value is dropped and the
// pseudo-inverse IS the inverse (inv_matrix.py:7-12 keeps s > 1e-15), log det = sum of the kept log s.  The certificate also
// asks for ||K||_F ||K^-1||_F <= cond_max, so that the computed inverse, and with it the bound, can be trusted.
struct CholCert { double fro_k, fro_inv, logdet; int ok, info; };
__global__ __launch_bounds__(1024) void k_cert_finish(const double* __restrict__ part_k, const double* __restrict__ part_inv,
                                                      const double* __restrict__ ldp, int nblk, const int* __restrict__ info,
                                                      double thresh, double cond_max, double* __restrict__ res,
                                                      CholCert* __restrict__ out) {
    __shared__ double red[2][16];
    double a = part_k[threadIdx.x], b = part_inv[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0., tb = 0., lg = 0.;
        for (int i = 0; i < 16; ++i) { ta += red[0][i]; tb += red[1][i]; }
        for (int i = 0; i < nblk; ++i) lg += ldp[i];
        const double fk = sqrt(ta), fi = sqrt(tb);
        // (every comparison is false on a NaN: a failed factorisation cannot pass)
        const int ok = info[0] == 0 && fi > 0. && fk * fi <= cond_max && 1. / fi > 100. * thresh;
        out->fro_k = fk; out->fro_inv = fi; out->logdet = lg; out->ok = ok; out->info = info[0];
        if (ok) { res[0] = lg; res[1] = 0.; }
    }
}

// A (rp x rp, lower triangle from the GEMM) before a Cholesky-QR pass: the padding diagonal (indices >= r) := 1 (X has zero
// columns there), the diagonal += shift * trace(A) over the r real indices
__global__ __launch_bounds__(256) void k_cholqr_prepare(double* __restrict__ A, long ld, int r, int rp, double shift) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double t = 0.;
    for (int i = tid; i < r; i += 256) t += A[(long)i * ld + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    const double s = shift * ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = tid; i < rp; i += 256) {
        if (i < r) A[(long)i * ld + i] += s;
        else A[(long)i * ld + i] = 1.;
    }
}
