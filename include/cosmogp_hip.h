/*
 * cosmogp_hip.h -- C ABI of libcosmogp_hip.so, the MI355X (gfx950) implementation of
 * the cosmogp GP hot path: RBF covariance assembly + blocked Cholesky factor/solve.
 *
 * This is the drop-in boundary.  The reference (PFLeget/cosmogp) has no FFI of its own;
 * its plugin seam is:
 *   - two module-level callables bound at import, cosmogp/Gaussian_process.py:6-9
 *       svd(matrix, return_logdet=False) / chol(matrix, return_logdet=False)
 *     (satisfied by an optional external C++ binding `svd_tmv` or by
 *     cosmogp/inv_matrix.py:4-31), and
 *   - the kernel callable bound per instance by name, cosmogp/Gaussian_process.py:140-154
 *       kernel(x, hyperparameter, new_x=None, nugget=0., floor=0., y_err=None) -> ndarray
 *     implemented in cosmogp/kernel.py:25-155.
 * The entry points below are what a ctypes binding for that seam calls; the operator-level
 * functions mirror the seam 1:1, the fused-level functions keep the N x N matrices on the
 * device for the callers in cosmogp/Gaussian_process.py (log_likelihood_gp :13-75,
 * get_prediction :270-338, get_covariance_matrix :340-361).
 *
 * Conventions
 *   - all pointers are HOST pointers to C-contiguous float64 / int64 unless a name starts with d_
 *   - inputs are never modified (the reference calls scipy with overwrite_a=False,
 *     cosmogp/inv_matrix.py:23); outputs are caller-allocated
 *   - return value: 0 = ok; > 0 = LAPACK-style info (1-based index of the first
 *     non-positive pivot: the reference raises numpy.linalg.LinAlgError there,
 *     cosmogp/inv_matrix.py:23); < 0 = CGP_E_* (argument / HIP error, text via cgp_last_error)
 *   - thread-compatible: one handle per thread; each handle owns one HIP stream and its
 *     device workspaces; calls on one handle are serialised by the caller
 *   - no torch / numpy types cross this boundary.
 *
 * Kernel parameters ("kp", 7 doubles), computed by the host exactly as the reference does:
 *   kp[0] amp_train  multiplies exp() in the train-train matrix
 *                    1D: h0**2 (kernel.py:72); 2D: 1.0 -- the reference omits sigma^2 there
 *                    (kernel.py:146) -- or h0**2 in corrected mode
 *   kp[1] amp_cross  multiplies exp() in the cross matrix: h0**2 (kernel.py:72, :141)
 *   kp[2..5]         1D: kp[2] = h1**2 (kernel.py:72)
 *                    2D: VI = inv_l row-major [VI00, VI01, VI10, VI11] (kernel.py:127-130)
 *   kp[6]            nug2: added on the device to every element of `diag` (0 when unused)
 * The diagonal term (y_err**2 + floor**2 + nugget**2, kernel.py:74-75/:150-151) is
 * diag[i] + kp[6] with diag = y_err**2 + floor**2 a length-n vector evaluated by the host: the
 * reference's left-to-right order (a + b) + c is kept, and an objective evaluation with a new nugget
 * uploads 8 bytes instead of a vector.  (Callers may also fold everything into `diag`, kp[6] = 0.)
 */
#ifndef COSMOGP_HIP_H
#define COSMOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cgp_handle cgp_handle;

enum {
    CGP_OK = 0,
    CGP_E_ARG = -1,      /* bad argument */
    CGP_E_HIP = -2,      /* HIP runtime error */
    CGP_E_STATE = -3,    /* call order (e.g. predict before factor) */
    CGP_E_NOMEM = -4
};

enum { CGP_DIM_1D = 1, CGP_DIM_2D = 2 };

/* which matrix cgp_fit_fetch returns */
enum { CGP_MAT_K = 0, CGP_MAT_L = 1, CGP_MAT_KINV = 2 };

/* ---- library / device ------------------------------------------------------------- */
int cgp_version(void);                         /* ABI version, currently 1 */
int cgp_device_count(int* count);
int cgp_create(int device, cgp_handle** out);
int cgp_destroy(cgp_handle* h);
const char* cgp_last_error(cgp_handle* h);     /* text of the last <0 status ("" if none) */
int cgp_synchronize(cgp_handle* h);

/* ---- operator level: mirrors the reference seam 1:1 -------------------------------- */

/* kernel(x, hyp, new_x, nugget, floor, y_err)  -- cosmogp/kernel.py:25-77 (dim 1), :80-155 (dim 2).
 * x: n points (n doubles, or n x 2 AoS).  xnew == NULL: train-train, out is (n, n) with `diag`
 * (n doubles, may be NULL = zeros) added on the diagonal.  xnew != NULL: cross matrix, out is
 * (m, n), rows = new points, no diagonal term.  ld = row stride of out in doubles (>= n). */
int cgp_rbf(cgp_handle* h, int dim, const double* x, int64_t n,
            const double* xnew, int64_t m, const double* kp, const double* diag,
            double* out, int64_t ld);

/* chol(matrix, return_logdet=True) -- cosmogp/inv_matrix.py:21-31.  K: (n, n) row stride ldk,
 * symmetric positive definite (lower triangle is read).  inv_out: (n, n) dense, row stride n.
 * logdet_out may be NULL.  Returns info > 0 when K is not positive definite. */
int cgp_cholesky_inverse(cgp_handle* h, const double* K, int64_t n, int64_t ldk,
                         double* inv_out, double* logdet_out);

/* svd(matrix, return_logdet=True) -- cosmogp/inv_matrix.py:4-18: pseudo-inverse of a SYMMETRIC matrix
 * with the modes whose singular value is <= thresh (reference: 1e-15) dropped; logdet = sum of the logs
 * of the kept singular values; never fails on singular input.  On the device the SVD of the symmetric K
 * is its eigen-decomposition (two-sided block Jacobi, csrc/jacobi.hpp): s = |lambda|.
 * K: (n, n) row stride ldk, both triangles read.  inv_out: (n, n) dense.  n_dropped may be NULL.
 * EFFECTIVE THRESHOLD of the low-rank pre-stage (route 1 of cgp_eig_info; matrices of n >= 256 that are numerically rank
 * deficient): the pivoted Cholesky K = G G^T + E stops at a residual diagonal of 4 eps max K_ii and is accepted when
 * |E|_F <= 8 n eps max K_ii, so modes below that bound are dropped whatever `thresh` says: the route behaves like
 * thresh_eff = max(thresh, 8 n eps max K_ii).  For a threshold above that bound the routes agree (tests/test_eigen.py);
 * at the reference's absolute 1e-15 on a matrix of scale ~1 the full sweeps (route 0) keep the rounding-noise modes between
 * 1e-15 and the bound (each contributes ~ -30 to logdet and 1 / lambda ~ 1e13..1e15 to the pseudo-inverse), route 1 drops
 * them: n_dropped and logdet differ between the routes there (no decomposition reproduces those modes: DESIGN.md 4.5);
 * CGP_JAC_NO_LOWRANK=1 forces route 0. */
int cgp_svd_inverse(cgp_handle* h, const double* K, int64_t n, int64_t ldk, double thresh,
                    double* inv_out, double* logdet_out, int64_t* n_dropped);

/* Diagnostic: which way the last eigen-route call of this handle (cgp_svd_inverse, cgp_fit_svd) went.
 *   out[0]  0 = Jacobi sweeps on the whole matrix; 1 = low-rank pre-stage (pivoted Cholesky + Rayleigh-Ritz: numerically
 *           rank-deficient, i.e. noise-free, matrices); 2 = certified Cholesky (positive definite, well conditioned and
 *           1 / |K^-1|_F above the threshold: no mode can be dropped, the inverse is the pseudo-inverse)
 *   out[1]  rank found by the pre-stage (-1: none);  out[2]  outer Jacobi sweeps run (0 for route 2) */
int cgp_eig_info(cgp_handle* h, int64_t out[3]);

/* ---- fused level: one object's fit, matrices stay in HBM ---------------------------- */

/* Upload one object: coordinates x (n or n x 2) and residual r = y - y0 (n doubles).
 * Replaces any previous object on this handle. */
int cgp_fit_set_data(cgp_handle* h, int dim, const double* x, const double* resid, int64_t n);

/* Assemble K(kp, diag) on the device, factor it (lower Cholesky, in place) and solve the
 * residual row; optionally, in the same pass, assemble the cross matrix for m new points,
 * solve it and produce the prediction.
 *   diag      n doubles: uploaded and kept resident; NULL = keep the resident vector (zeros right
 *             after cgp_fit_set_data)
 *   xnew      m points, or NULL (m = 0): log-likelihood only
 *   mean_new  m doubles added to the predicted mean (NULL = 0)       [Gaussian_process.py:335]
 *   kss_noise scalar diagonal term of K(new,new): (0**2 + floor**2 + nugget**2); the device
 *             adds it to amp_train like kernel.py:74-75 does                [Gaussian_process.py:357]
 *   loglik    -1/2 r'K^-1 r - n/2 log(2 pi) - 1/2 logdet             [Gaussian_process.py:68-73]
 *   mu        m doubles: H K^-1 r + mean_new                          [Gaussian_process.py:334]
 *   var       m doubles or NULL: diag(K** - H K^-1 H')               [Gaussian_process.py:359-361]
 * Returns info > 0 if K is not positive definite (outputs are then NaN). */
int cgp_fit(cgp_handle* h, const double* kp, const double* diag,
            const double* xnew, int64_t m, const double* mean_new, double kss_noise,
            double* loglik, double* logdet, double* mu, double* var);

/* The same fit by the reference's svd_method=True algorithm (Gaussian_process.py:54-57, :319-320): K is
 * assembled in HBM, eigen-decomposed there, K^+ replaces K^-1 in the likelihood (:68-73), the prediction
 * (:334-335) and the predictive covariance (:356-361).  cov: NULL or (m, m) dense.  Invalidates the
 * Cholesky factor of the handle.  Returns 0 also for singular K (modes <= thresh are dropped). */
int cgp_fit_svd(cgp_handle* h, const double* kp, const double* diag,
                const double* xnew, int64_t m, const double* mean_new, double kss_noise, double thresh,
                double* loglik, double* logdet, int64_t* n_dropped, double* mu, double* var, double* cov);

/* Prediction on further points with the factor left by the last cgp_fit (same kp).
 * cov: NULL or (m, m) dense = K(new,new; nugget) - H K^-1 H'  [Gaussian_process.py:356-361] */
int cgp_fit_predict(cgp_handle* h, const double* xnew, int64_t m, const double* mean_new,
                    double kss_noise, double* mu, double* var, double* cov);

/* Log-likelihood of the resident object AND the moment sums of its analytic gradient
 * (gradient-based alternative to the derivative-free scipy.optimize.fmin of
 * Gaussian_process.py:247; the likelihood is the one of Gaussian_process.py:68-73).  With
 * K_ij = kp[0] E_ij + delta_ij (diag_i + kp[6]), E = exp(-q/2), M = alpha alpha' - K^-1:
 *   sums[0] = sum M_ij E_ij          sums[1] = sum M_ij E_ij d0^2    sums[2] = sum M_ij E_ij d0 d1
 *   sums[3] = sum M_ij E_ij d1^2     sums[4] = sum_i M_ii            (d = x_j - x_i; 1D: d1 = 0)
 * so that dlogL/dkp[0] = sums[0]/2, dlogL/dkp[6] = sums[4]/2, 1D: dlogL/dkp[2] = kp[0] sums[1] / (4 kp[2]^2),
 * 2D: dlogL/dkp[2..5] = -kp[0]/4 * (sums[1], sums[2], sums[2], sums[3]).  The caller applies the chain
 * rule to its own hyperparameters.  Returns like cgp_fit (> 0: not positive definite). */
int cgp_fit_grad(cgp_handle* h, const double* kp, const double* diag,
                 double* loglik, double* logdet, double* sums);

/* Lazy attributes of the reference object (kernel_matrix :262, inv_kernel_matrix :325):
 * which = CGP_MAT_K (re-assembled, symmetric), CGP_MAT_L (lower factor, zeros above),
 * CGP_MAT_KINV (explicit inverse from the factor).  out: (n, n) dense. */
int cgp_fit_fetch(cgp_handle* h, int which, double* out);

/* ---- batched: B independent objects of (possibly ragged) size, one launch sequence ---- */

/* ns[B] sizes; x / resid / diag packed object after object (offsets = prefix sums of ns,
 * times dim for x).  xnew: m points -- one grid shared by all objects (xnew_per_object = 0,
 * the new_binning of Gaussian_process.py:287-289) or B grids of m points each
 * (xnew_per_object = 1; leave-one-out of cosmogp/pull.py:75-90) -- may be NULL (m = 0).
 * mean_new: m doubles shared by all objects or NULL.
 * Outputs: loglik[B], info[B] (0 or first bad pivot), mu[B*m], var[B*m] (may be NULL).
 * B <= 65535 per call (the object index is a grid dimension): CGP_E_ARG beyond; the Python classes cut longer
 * lists into several calls (gaussian_process.MAX_BATCH_OBJECTS). */
int cgp_batch_set_data(cgp_handle* h, int dim, int64_t B, const int64_t* ns,
                       const double* x, const double* resid, const double* diag);
/* x == resid == NULL: fit the batch left resident by cgp_batch_set_data (B, dim must match; ns, diag
 * ignored) -- the form the fit driver uses, one call per objective evaluation. */
int cgp_batch_fit(cgp_handle* h, int dim, int64_t B, const int64_t* ns,
                  const double* x, const double* resid, const double* diag,
                  const double* kp, const double* xnew, int64_t m, int xnew_per_object,
                  const double* mean_new, double kss_noise, double* loglik, int32_t* info,
                  double* mu, double* var);

/* cgp_fit_grad for every object of the RESIDENT batch (cgp_batch_set_data) in one launch sequence, one
 * kernel-parameter set shared by the objects: loglik[B], info[B] (> 0: object not positive definite),
 * sums[B][5] as in cgp_fit_grad. */
int cgp_batch_fit_grad(cgp_handle* h, const double* kp, double* loglik, int32_t* info, double* sums);

/* Hyper-parameter sweep (BASELINE config[4]; the loop a caller would write around
 * Gaussian_process.compute_log_likelihood, Gaussian_process.py:191-213): log-likelihood of the RESIDENT
 * object (cgp_fit_set_data) for np kernel-parameter sets kp[np][7] in ONE launch sequence -- np matrices in
 * HBM, coordinates / residual / diagonal shared.  np <= 256 per call; info[c] > 0: candidate c is not positive
 * definite (its loglik is then meaningless).  Replaces the resident batch of the handle. */
int cgp_sweep_loglik(cgp_handle* h, int64_t np, const double* kp, const double* diag,
                     double* loglik, int32_t* info);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------
 * The units of this path (objects of Gaussian_process.py:207-211 / :304-335, hyper-parameter candidates of a
 * sweep) are independent: every rank fits its own contiguous block on its own GPU, nothing is exchanged while
 * fitting, and ONE ncclAllGather of the per-unit results ends the call.  The communicator belongs to the
 * handle and runs on the handle's stream; librccl.so.1 is opened on first use (the single-GPU path never
 * loads it).  Bootstrap: rank 0 calls cgp_comm_unique_id and hands the 128 bytes to the other ranks by any
 * out-of-band channel (cosmogp_amd/multigpu.py uses a file or a TCP socket on 127.0.0.1), then every rank
 * calls cgp_comm_init. */
#define CGP_COMM_ID_BYTES 128
enum { CGP_OP_SUM = 0, CGP_OP_MAX = 1 };
int cgp_comm_unique_id(void* id_out);                                  /* CGP_COMM_ID_BYTES bytes */
int cgp_comm_init(cgp_handle* h, int rank, int world, const void* id); /* collective over all ranks */
int cgp_comm_destroy(cgp_handle* h);
/* ranks the communicator itself reports (ncclCommCount): what a caller records to show that RCCL really spans `world` ranks */
int cgp_comm_count(cgp_handle* h, int* ranks);
/* recv[world][count] <- send[count] of every rank (host pointers, staged through HBM, ncclAllGather) */
int cgp_comm_allgather(cgp_handle* h, const double* send, int64_t count, double* recv);
/* buf[count] <- elementwise sum / max over ranks (ncclAllReduce); count = 0 is a barrier */
int cgp_comm_allreduce(cgp_handle* h, double* buf, int64_t count, int op);
/* cgp_sweep_loglik over the ranks of the communicator (BASELINE config[4]): kp[np_total][7] is the FULL
 * candidate table on every rank; rank r evaluates the contiguous block [lo_r, hi_r) (sizes differ by at most
 * one) in lock-step launch sequences of at most `chunk` candidates and the log-likelihoods / info words are
 * gathered device to device (no host hop before the collective): loglik[np_total], info[np_total] complete
 * on every rank.  The object must be resident on every rank (cgp_fit_set_data). */
int cgp_sweep_loglik_sharded(cgp_handle* h, int64_t np_total, const double* kp, const double* diag,
                             int64_t chunk, double* loglik, int32_t* info);

/* ---- leave-one-out ingredients (caller of the path) ----------------------------------
 * cosmogp/pull.py:51-90 builds a GP on n - 1 points and predicts the left-out one, n times per object.  With the factor of
 * the FULL matrix K (all n points, same kp)
 *     k_t' K_-t^-1 v_-t = v_t - (K^-1 v)_t / (K^-1)_tt ,     k_t' K_-t^-1 k_t = K_tt - 1 / (K^-1)_tt
 * so alpha = K^-1 r, beta = K^-1 1 and g = diag(K^-1) give every leave-one-out mean (beta: for a mean offset that depends
 * on the left-out point, mean.py:95-97) and variance: one factorisation per object instead of n.  Runs on the resident
 * batch (cgp_batch_set_data); alpha, beta, g: [B][Np] doubles, Np = the batch's padded size (max n rounded up to 128),
 * entries t >= n_b are not written; info[b] as cgp_batch_fit. */
int cgp_batch_loo(cgp_handle* h, const double* kp, double* alpha, double* beta, double* g, int32_t* info);

/* ---- mean function at the edge of the path ---------------------------------------
 * interpolate_mean_2d, cosmogp/mean.py:36-67, evaluates a bivariate spline with one `bisplev` call per point
 * (:63-65).  cgp_bispev evaluates the tensor-product B-spline (tx[nx], ty[ny], c[(nx-kx-1)*(ny-ky-1)], degrees kx,
 * ky <= 5: the `tck` of scipy.interpolate.bisplrep, fitted on the host) at m points pts[m][2] in one launch, FITPACK's
 * fpbisp / fpbspl arithmetic (arguments outside the base interval take the boundary value).  out[m]. */
int cgp_bispev(cgp_handle* h, const double* tx, int64_t nx, const double* ty, int64_t ny, const double* c, int kx, int ky,
               const double* pts, int64_t m, double* out);

/* ---- measurement hooks (bench.py / tests only) ------------------------------------- */

/* Device-resident timing of the last cgp_fit / cgp_batch_fit: runs `iters` fits (or batches) back to back with
 * inputs already in HBM and no host copies, returns average milliseconds per fit measured
 * with HIP events on the handle's stream.  stage_ms[12] (may be NULL) receives per-stage
 * averages from a second, event-instrumented pass:
 *   [0] assemble K  [1] potrf_diag  [2] trsm  [3] gemm update  [4] assemble cross
 *   [5] finish (mu/var/loglik)  [6] count of gemm-update launches (of k_chol_dag launches when [11] > 0)  [7] gemm flops per fit
 *   [8] the part of [3] spent in the 128x64-tile form  [9] its launches  [10] its flops
 *   [11] the single-launch factorisation (csrc/dag.hpp: one object, or a batch of objects with >= 16 tile columns, 4-16
 *        objects per launch; [1]-[3] are then 0)
 * stage_ms must hold 12 doubles. */
int cgp_bench_fit(cgp_handle* h, int iters, double* ms_per_fit, double* stage_ms);

/* Same for the resident BATCH (after cgp_batch_fit): `iters` passes over its first nobj objects
 * (1 <= nobj <= B), all in one launch sequence; ms_per_pass = time of one pass. */
int cgp_bench_fit_objects(cgp_handle* h, int iters, int64_t nobj, double* ms_per_pass);

/* Bare v_mfma_f64_4x4x4_4b_f64 issue loop on every CU: measured fp64 matrix ceiling (TFLOP/s),
 * and a streaming-store kernel: measured HBM write ceiling (GB/s). */
int cgp_microbench(cgp_handle* h, double* mfma_f64_tflops, double* hbm_write_gbs);

/* Checks the f64 MFMA operand / result lane maps the kernels rely on with exact integer
 * data (asymmetric B).  Returns 0 when they hold. */
int cgp_selftest_mfma(cgp_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* COSMOGP_HIP_H */
