// Device kernels of libcosmogp_hip (gfx950 / CDNA4 only, wave64, fp64).
//
// Data layout in HBM: one fit owns a row-major matrix buffer with
// leading dimension ld = Np (n rounded up to NB = 128):
//     rows [0, Np)      K -> L   lower triangle, padded with the identity
//     rows [Np, R)      "tall" rows: cross matrix H (one row per new point), then the
//                        residual row r; the blocked factorisation turns them into
//                        W = H L^-T and z = L^-1 r in the same sweep (no separate solves)
// Batched problems are the same buffers with a batch stride (blockIdx.z).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rbfmath.hpp"

#define CGP_NB 128          // leaf block of the Cholesky (diagonal block lives in LDS)
#define CGP_SB 16           // sub-block inside the leaf (pivot block of K3, column tile of K4)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// In-kernel phase stamps: diagnostic builds only (tools/leaf_stamps.hip defines CGP_STAMPS).
#ifdef CGP_STAMPS
__device__ unsigned long long g_stamps[256];
#define CGP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_stamps[i] = t_; } } while (0)
#else
#define CGP_STAMP(i) do {} while (0)
#endif

struct KParams {            // mirrors the 7-double "kp" of include/cosmogp_hip.h ...
    double amp_train, amp_cross, p0, p1, p2, p3;
    double nug2;            // added to every element of the resident diagonal vector: the nugget**2 term
                            // of (y_err**2 + floor**2) + nugget**2 (kernel.py:75), so that a new nugget
                            // costs 8 bytes of upload instead of a new diagonal vector
    double kss_noise;       // the K(new,new) diagonal term; in HBM (like everything above) so that a
                            // captured graph is replayed with new hyper-parameters without re-capturing
};

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction.  Measured on MI355X
// (tools/microbench_f64.hip, tools/mfma4_layout.hip): 16-17 cycles issue (~30 flop/clk/SIMD,
// 68 TFLOP/s chip-wide) against ~105 cycles for the 16x16x4 form (~47 TFLOP/s), dependent
// latency 44 cycles.  Lane maps (probed with unit impulses):
//   A: lane l holds A_blk[i = l & 3][k = l >> 4],  blk = (l >> 2) & 3
//   B: lane l holds B_blk[k = l >> 4][j = l & 3]
//   D: lane l holds D_blk[i = l >> 4][j = l & 3]      (so D_blk is again a B operand of blk)
// (The 16x16x4 form, whose D layout differs from every other dtype -- cdna_hip_programming.md 3 -- is
// not used: it sustains only ~47 TFLOP/s.)  cgp_selftest_mfma re-checks these maps with exact integers.
__device__ __forceinline__ double mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// Same-wave LDS hand-off (one lane writes, other lanes of the SAME wave read): the LDS pipe is
// in order per wave, so no barrier is needed, but the compiler must not reorder the accesses
// (stores as double and loads as d2 do not alias for its type-based analysis).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------
// K1 / K2: covariance assembly.  Arithmetic follows the reference operation by operation
// (cosmogp/kernel.py:71-72 for 1D; scipy 'mahalanobis' + kernel.py:140-146 for 2D) with FP
// contraction off and every operation rounded like the reference's, so the exp() argument is
// bit-identical to NumPy's; the division and the square root are the correctly rounded ones by the
// short sequences of rbfmath.hpp (checked against the IEEE operations on the host,
// tests/native/check_rbfmath.cpp), exp is rbfmath's table-based one (< 1 ulp): only it differs from
// libm, by <= 1 ulp.  (The IEEE library forms of the three cost ~80 issue slots per element and
// made the kernel VALU-bound at 0.36 of the HBM roof: tools/asm_ablate.hip.)
// ------------------------------------------------------------------------------------
__device__ const double g_exp_tab[128] = RBF_EXP_TABLE_INIT;

// per-thread constants of the argument: 1D the correctly rounded reciprocal of the squared length scale
template <int DIM>
__device__ __forceinline__ double rbf_arg_const(const KParams& kp) {
    return DIM == 1 ? 1.0 / kp.p0 : 0.;
}

// DIAGVI (2D): lxy = 0, the off-diagonal entries of VI are zero and their products exact zeros: 4 operations less, same bits
template <int DIM, bool DIAGVI = false>
__device__ __forceinline__ double rbf_exp_arg(double ux, double uy, double vx, double vy,
                                              const KParams& kp, double yinv) {
#pragma clang fp contract(off)
    if (DIM == 1) {
        double d = ux - vx;
        return -0.5 * rbf_div(d * d, kp.p0, yinv);
    } else {
        double d0 = ux - vx, d1 = uy - vy;
        double t0, t1;
        if (DIAGVI) {
            t0 = d0 * kp.p0;
            t1 = d1 * kp.p3;
        } else {
            t0 = d0 * kp.p0 + d1 * kp.p1;
            t1 = d0 * kp.p2 + d1 * kp.p3;
        }
        double s = d0 * t0 + d1 * t1;
        double dist = rbf_sqrt_arg(s);
        return -0.5 * (dist * dist);
    }
}

// the 2^(j/64) table of rbf_exp into LDS (callers: before their first __syncthreads())
__device__ __forceinline__ void rbf_stage_table(double* etab) {
    for (int t = threadIdx.x; t < 128; t += blockDim.x) etab[t] = g_exp_tab[t];
}

// Train-train block: out[i][j] for i, j in [0, Np); 64 x 128 tile per workgroup, each lane
// owns two adjacent columns (16-byte stores, 1 KiB per wave store = full lines), the 64 row
// points of the tile are staged in LDS and read as broadcasts.
// lower_only: tiles entirely above the diagonal are skipped (the factorisation reads the
// lower triangle only).  Indices >= n (padding) get the identity.
template <int DIM>
__global__ __launch_bounds__(256) void k_assemble_train(
    double* __restrict__ M, long ld, long sM, const double* __restrict__ x, long sx,
    const double* __restrict__ diag, long sd, const int* __restrict__ ns, const KParams* __restrict__ kpp,
    int lower_only, int use_nug2, int kps = 0) {
    const int b = blockIdx.z;
    const int col0 = blockIdx.x * 128, row0 = blockIdx.y * 64;
    if (lower_only && col0 > row0 + 63) return;
    const int n = ns[b];
    M += (long)b * sM;
    x += (long)b * sx;
    diag += (long)b * sd;
    const KParams kp = kpp[(long)b * kps];          // kps = 1: one parameter set per object (hyper-parameter sweeps)
    const double yinv = rbf_arg_const<DIM>(kp);
    __shared__ double rx[64 * 2];
    __shared__ __attribute__((aligned(16))) double etab[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64 * DIM) rx[tid] = x[(long)row0 * DIM + tid];
    rbf_stage_table(etab);
    const int c = col0 + 2 * lane;
    double ux0, uy0 = 0., ux1, uy1 = 0.;
    if (DIM == 1) {
        d2 t = *(const d2*)(x + c);
        ux0 = t.x; ux1 = t.y;
    } else {
        d2 t0 = *(const d2*)(x + 2 * (long)c), t1 = *(const d2*)(x + 2 * (long)c + 2);
        ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y;
    }
    __syncthreads();
    if (row0 + 63 < n && col0 + 127 < n && (col0 + 127 < row0 || col0 > row0 + 63)) {
        // interior tile (no diagonal element, no padding): no selects
        auto rows = [&](auto dg) {
            constexpr bool DG = decltype(dg)::value;
#pragma unroll 4
            for (int rr = wave; rr < 64; rr += 4) {
                const int i = row0 + rr;
                const double vx = rx[rr * DIM], vy = (DIM == 2) ? rx[rr * DIM + 1] : 0.;
                d2 o;
                {
#pragma clang fp contract(off)
                    o.x = kp.amp_train * rbf_exp(rbf_exp_arg<DIM, DG>(ux0, uy0, vx, vy, kp, yinv), etab);
                    o.y = kp.amp_train * rbf_exp(rbf_exp_arg<DIM, DG>(ux1, uy1, vx, vy, kp, yinv), etab);
                }
                *(d2*)(M + (long)i * ld + c) = o;
            }
        };
        if (DIM == 2 && kp.p1 == 0. && kp.p2 == 0.) rows(std::true_type{});
        else rows(std::false_type{});
        return;
    }
#pragma unroll 4
    for (int rr = wave; rr < 64; rr += 4) {
        const int i = row0 + rr;
        double vx = rx[rr * DIM], vy = (DIM == 2) ? rx[rr * DIM + 1] : 0.;
        d2 o;
        {
#pragma clang fp contract(off)
            double e0 = kp.amp_train * rbf_exp(rbf_exp_arg<DIM>(ux0, uy0, vx, vy, kp, yinv), etab);
            double e1 = kp.amp_train * rbf_exp(rbf_exp_arg<DIM>(ux1, uy1, vx, vy, kp, yinv), etab);
            if (i < n) {
                double dg = kp.amp_train + (use_nug2 ? (diag[i] + kp.nug2) : diag[i]);
                o.x = (c < n) ? ((c == i) ? dg : e0) : 0.;
                o.y = (c + 1 < n) ? ((c + 1 == i) ? dg : e1) : 0.;
            } else {
                o.x = (c == i) ? 1. : 0.;
                o.y = (c + 1 == i) ? 1. : 0.;
            }
        }
        *(d2*)(M + (long)i * ld + c) = o;
    }
}

// Cross block: rows t in [0, Mp) of the tall part.  t < m: amp_cross * exp(arg(x_j, xnew_t));
// t >= m: zeros (the residual row is written separately into the last tall row).
template <int DIM>
__global__ __launch_bounds__(256) void k_assemble_cross(
    double* __restrict__ M, long ld, long sM, const double* __restrict__ x, long sx,
    const double* __restrict__ xnew, long sxn, const int* __restrict__ ns,
    const KParams* __restrict__ kpp, int m, int kps = 0) {
    const int b = blockIdx.z;
    const int col0 = blockIdx.x * 128, row0 = blockIdx.y * 64;
    const int n = ns[b];
    M += (long)b * sM;
    x += (long)b * sx;
    xnew += (long)b * sxn;
    const KParams kp = kpp[(long)b * kps];
    const double yinv = rbf_arg_const<DIM>(kp);
    __shared__ double rx[64 * 2];
    __shared__ __attribute__((aligned(16))) double etab[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64 * DIM) rx[tid] = xnew[(long)row0 * DIM + tid];
    rbf_stage_table(etab);
    const int c = col0 + 2 * lane;
    double ux0, uy0 = 0., ux1, uy1 = 0.;
    if (DIM == 1) {
        d2 t = *(const d2*)(x + c);
        ux0 = t.x; ux1 = t.y;
    } else {
        d2 t0 = *(const d2*)(x + 2 * (long)c), t1 = *(const d2*)(x + 2 * (long)c + 2);
        ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = wave; rr < 64; rr += 4) {
        const int t = row0 + rr;
        d2 o;
        o.x = 0.; o.y = 0.;
        if (t < m) {
#pragma clang fp contract(off)
            double vx = rx[rr * DIM], vy = (DIM == 2) ? rx[rr * DIM + 1] : 0.;
            double e0 = kp.amp_cross * rbf_exp(rbf_exp_arg<DIM>(ux0, uy0, vx, vy, kp, yinv), etab);
            double e1 = kp.amp_cross * rbf_exp(rbf_exp_arg<DIM>(ux1, uy1, vx, vy, kp, yinv), etab);
            o.x = (c < n) ? e0 : 0.;
            o.y = (c + 1 < n) ? e1 : 0.;
        }
        *(d2*)(M + (long)t * ld + c) = o;
    }
}

// rows [row0, row0+nrows) := identity shifted so that row row0+i has its 1 in column i
// (tall rows for the explicit inverse W = I L^-T)
__global__ void k_set_identity_rows(double* __restrict__ M, long ld, int nrows, long sM = 0) {
    M += (long)blockIdx.z * sM;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tot = (long)nrows * ld;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        long r = idx / ld, c = idx - r * ld;
        M[idx] = (r == c) ? 1. : 0.;
    }
}

// out (n x n dense) := lower triangle of M mirrored (sym = 1) or with zeros above (sym = 0)
__global__ void k_extract_square(const double* __restrict__ M, long ld, double* __restrict__ out,
                                 int n, int mode /*0 = as is, 1 = lower+zeros, 2 = mirror lower*/) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tot = (long)n * n;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        long r = idx / n, c = idx - r * n;
        double v;
        if (mode == 0) v = M[r * ld + c];
        else if (mode == 1) v = (c <= r) ? M[r * ld + c] : 0.;
        else v = (c <= r) ? M[r * ld + c] : M[c * ld + r];
        out[idx] = v;
    }
}

// pad an (n x n, row stride lds) host-uploaded matrix into the Np-padded buffer (identity padding)
__global__ void k_pad_square(const double* __restrict__ src, long lds, double* __restrict__ M,
                             long ld, int n, int Np) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tot = (long)Np * Np;
    for (; idx < tot; idx += (long)gridDim.x * blockDim.x) {
        long r = idx / Np, c = idx - r * Np;
        double v = (r < n && c < n) ? src[r * lds + c] : ((r == c) ? 1. : 0.);
        M[r * ld + c] = v;
    }
}

// ------------------------------------------------------------------------------------
// First-touch generation (round 3; objects whose whole fit runs on the launch-per-step route with the small update tiles:
// R < 2048, config 4).  A fit of such an object is HBM-bound -- 8.9 MB of traffic for 0.07 Gflop at N = 512 -- and 3.1 MB of
// it is the assembled matrix written by K1 / K2 and read back by whichever kernel touches a tile first.  Those kernels are
// known: tile (0, 0) is first read by the factorisation of block column 0, the rest of block column 0 by its panel solve,
// every other block column by the ONE update launch with c0 = 0 that covers it (factor_range's left spine).  With GEN != 0
// they compute the entry instead of loading it (the same device functions as K1 / K2, operation by operation: same bits),
// and the assembly launches disappear.  GEN = 1 / 2: coordinates of 1 / 2 components.
//   rows < Np: train rows (index >= n: identity padding); Np + t, t < m: cross row of new point t; zrow: the residual;
//   every other tall row: zeros.
// ------------------------------------------------------------------------------------
struct GenSrc {
    const double* x = nullptr; long sx = 0;
    const double* xnew = nullptr; long sxn = 0;
    const double* diag = nullptr; long sd = 0;
    const double* resid = nullptr; long sr = 0;
    const int* ns = nullptr;
    const KParams* kpp = nullptr; int kps = 0;
    int m = 0, Np = 0, zrow = 0;
    int row0 = 0, col0 = 0;          // absolute (row, column) of the origin of the launch's region
    // PACKED small objects (round 4): pack = 2 / 4 objects of <= hs = 64 / 32 points share one 128 x 128 tile as a block
    // diagonal (the Cholesky factor of a block-diagonal matrix is the block diagonal of the factors, and the products with
    // the zero blocks are exact zeros: every object's numbers are those of its own tile, bit for bit).  Matrix row / column i
    // belongs to local object i / hs; the coordinate / diagonal / residual arrays hold hs entries per object, so a tile's 128
    // entries are its objects' arrays back to back; ns has `pack` entries per tile; tall rows: pack x m cross rows (object
    // lo: rows Np + lo m ..., shared grid only), zero rows, then one residual row per object at zrow + lo.  hs = 0: unpacked.
    int pack = 1, hs = 0;
    long sxo = 0;                    // packed tiles with one prediction grid PER OBJECT: stride of an object's grid inside its tile's block (0: shared grid)
};
struct GenObj { const double* x; const double* xnew; const double* diag; const double* resid; const int* ns; KParams kp; double yinv; int n; bool dvi; };
// kind 0 train row, 1 identity padding, 2 cross row, 3 zero row, 4 residual; the row's object owns columns [c0, c0 + n)
struct GenRow { double vx, vy, dg; int kind, c0, n; };

template <int DIM>
__device__ __forceinline__ GenObj gen_obj(const GenSrc& g, int b) {
    GenObj o;
    o.x = g.x + (long)b * g.sx; o.xnew = g.xnew + (long)b * g.sxn; o.diag = g.diag + (long)b * g.sd; o.resid = g.resid + (long)b * g.sr;
    o.kp = g.kpp[(long)b * g.kps];
    o.yinv = rbf_arg_const<DIM>(o.kp);
    o.ns = g.ns + (long)b * g.pack;
    o.n = o.ns[0];
    o.dvi = DIM == 2 && o.kp.p1 == 0. && o.kp.p2 == 0.;
    return o;
}
template <int DIM>
__device__ __forceinline__ GenRow gen_row(const GenSrc& g, const GenObj& o, int i) {
    GenRow r;
    r.vx = 0.; r.vy = 0.; r.dg = 0.;
    r.c0 = 0; r.n = o.n;
    if (i < g.Np) {
        int il = i;
        if (g.hs) { const int lo = i / g.hs; r.c0 = lo * g.hs; r.n = o.ns[lo]; il = i - r.c0; }
        r.kind = il < r.n ? 0 : 1;
        r.vx = o.x[(long)i * DIM];
        if (DIM == 2) r.vy = o.x[(long)i * DIM + 1];
        if (il < r.n) r.dg = o.kp.amp_train + (o.diag[i] + o.kp.nug2);
    } else if (i - g.Np < g.pack * g.m) {
        int t = i - g.Np;
        const double* xn = o.xnew;
        if (g.hs) { const int lo = t / g.m; r.c0 = lo * g.hs; r.n = o.ns[lo]; t -= lo * g.m; xn += (long)lo * g.sxo; }
        r.kind = 2;
        r.vx = xn[(long)t * DIM];
        if (DIM == 2) r.vy = xn[(long)t * DIM + 1];
    } else if (i >= g.zrow && i < g.zrow + g.pack) {
        r.kind = 4;
        if (g.hs) { const int lo = i - g.zrow; r.c0 = lo * g.hs; r.n = o.ns[lo]; }
    } else r.kind = 3;
    return r;
}
// entry (i, c) of the object's assembled matrix; (ux, uy) = the point of column c
template <int DIM>
__device__ __forceinline__ double gen_val(const GenObj& o, const GenRow& r, int i, int c, double ux, double uy, const double* etab) {
#pragma clang fp contract(off)
    if (r.kind == 1) return c == i ? 1. : 0.;
    if (r.kind == 3) return 0.;
    const bool mine = c >= r.c0 && c < r.c0 + r.n;           // (unpacked: c0 = 0, n = the object's length)
    if (r.kind == 4) return mine ? o.resid[c] : 0.;
    const double a = o.dvi ? rbf_exp_arg<DIM, true>(ux, uy, r.vx, r.vy, o.kp, o.yinv) : rbf_exp_arg<DIM, false>(ux, uy, r.vx, r.vy, o.kp, o.yinv);
    const double e = (r.kind == 0 ? o.kp.amp_train : o.kp.amp_cross) * rbf_exp(a, etab);
    if (!mine) return 0.;
    return (r.kind == 0 && c == i) ? r.dg : e;
}

// ------------------------------------------------------------------------------------
// XCD-affine object mapping of the batched launches of small objects (round 4).  Workgroups are dealt round robin over
// the 8 XCDs by their linear id (MI355X_MICROARCH.md, workgroup dispatch; cdna_hip_programming.md T1), and every XCD has
// its own L2: with the plain grid (tiles, 1, objects) the 5-14 tiles of one object's update land on 8 different L2s and
// each fetches the object's panels from HBM for itself (config 4: 213 MB per launch for 59 MB of distinct panels).  The
// launch is therefore made with the object count padded to a multiple of 8 and the linear id re-read as
//   group c = id % 8 (the blocks that share an XCD), slot p = id / 8  ->  object c + 8 (p / gx), tile p % gx:
// all tiles of an object run on ONE XCD, back to back, and the panels are fetched once.  A pure speed choice: the
// arithmetic of a tile does not depend on where it runs (same bits, tests/test_gpu_parity.py).
// Returns false for the padding workgroups (object >= B).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool xcd_affine(int& obj, int& bx, int B) {
    const int gx = gridDim.x;
    const int id = blockIdx.x + gx * blockIdx.z;
    const int c = id & 7, p = id >> 3;
    obj = c + 8 * (p / gx);
    bx = p % gx;
    return obj < B;
}

// ------------------------------------------------------------------------------------
// K3: Cholesky of one NB x NB diagonal block, entirely in LDS, one workgroup (4 waves).
//   8 sub-steps of 16 columns:
//   (a) wave 0 factors the 16 x 16 diagonal sub-block T in registers: lane r < 16 owns row r;
//       lanes 16..31 carry the rows of the identity through the same elimination and end up
//       holding T^-T (the "tall" trick at register level); pivots travel by v_readlane;
//       1/sqrt(pivot) is v_rsq_f64 + two Goldschmidt steps (a dependent chain of ~10 FMAs
//       instead of the ~60 instructions of an IEEE sqrt followed by an IEEE divide).
//   (c) sub-panel below T: P = A_sub * T^-T          (4-block MFMA, 16 instructions per tile)
//   (e) trailing update inside the block: C -= P_i P_j^T   (same)
// Outputs: L (lower, in place in HBM), the eight T^-T blocks (for the panel solve K4),
// sum(2 log L_ii) for the log-determinant, and LAPACK-style info.
// ------------------------------------------------------------------------------------
#define CGP_LDA 130   // LDS row stride in doubles (1040 B: rows stay 16-B aligned)

// Pivot arithmetic of the elimination.  Measured dependent latencies of fp64 VALU ops on gfx950
// (tools/lat_f64.hip): v_fma_f64 32, v_mul/v_add_f64 44, v_rsq_f64 ~52, v_readlane pair + use ~56
// cycles, so the pivot chain is written for minimum DEPTH: y = rsq(d) is refined in one shot,
//   1/sqrt(d) = y (1 + e/2 + 3 e^2/8),  e = 1 - d y^2,
// applied directly to the column value:  l = v y + (v y e) p.  V_RSQ_F64 is accurate to 2^29 ulp
// (|e| <= 2^-22), so the truncation error 5 e^3 / 16 is < 1e-20: no further term is needed.
__device__ __forceinline__ void pivot_scale(double d, double v, double& l, double& sd) {
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;                 // ~ sqrt(d)
    const double vy = v * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    l = __builtin_fma(vy * e, p, vy);
    sd = __builtin_fma(t * e, p, t);        // off the chain: only stored
}

// one 16x16 tile product on the 4-block MFMA: acc[J] (+)= sum_k A[row][k] * B[4J+u][k], k = 0..15.
// Arow points at A[16-row tile][k0] (row stride CGP_LDA), lane supplies its own row (l & 15);
// Brow points at B[16-col tile][k0].  NEG negates A.  Lane fields: u = l&3, q = l>>4.
template <bool NEG>
__device__ __forceinline__ void tile_mma_nt(double (&acc)[4], const double* Arow, const double* Brow,
                                            int lr, int u, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        d2 a = *(const d2*)(Arow + lr * CGP_LDA + 8 * t + 2 * q);
        if (NEG) a = -a;
        d2 b[4];
#pragma unroll
        for (int J = 0; J < 4; ++J) b[J] = *(const d2*)(Brow + (4 * J + u) * CGP_LDA + 8 * t + 2 * q);
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.x, b[J].x, acc[J]);
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.y, b[J].y, acc[J]);
    }
}

#define CGP_K3_THREADS 512
#define CGP_K3_WAVES (CGP_K3_THREADS / 64)

// The factorisation proper of a 128 x 128 block that already sits in LDS (As, row stride CGP_LDA; only the tiles on
// and below the block diagonal are read).  On return As holds L (diagonal tiles lower triangular, zeros above), Ws the
// eight T^-T blocks.  Called by all CGP_K3_THREADS threads of the workgroup; returns (in wave 0) the 1-based local
// index of the first non-positive pivot, 0 if none.  Shared by K3 and by the leaf lane of the single-launch
// factorisation (dag.hpp).
__device__ __forceinline__ int potrf128_lds(double* As, double* Ws, double* Lc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    int bad = 0;   // 1-based local index of the first non-positive pivot (wave 0 only)
    // Schedule of sub-step s (two workgroup barriers; everything that is not on the pivot chain runs
    // under phase (a) of the pivot wave):
    //   I1  wave 0 : (a)(s)  factor T = tile (s,s), produce T^-T
    //       others : trailing tiles of sub-step s-1, C[ti][tj] -= P_i P_j^T for ti >= tj >= s, (ti,tj) != (s,s)
    //   -- barrier --
    //   I2  wave 0 : (c0) P[s+1] = A[s+1][s] T^-T, then (e0) tile (s+1,s+1) -= P[s+1] P[s+1]^T
    //       others : (c) P[ti] = A[ti][s] T^-T for ti >= s+2
    //   -- barrier --
    for (int s = 0; s < 8; ++s) {
        const int o = s * 16;
        CGP_STAMP(2 + 4 * s);
        // workers of I1 = waves that do NOT share SIMD 0 with the pivot wave (waves go to SIMDs round
        // robin, so wave 4 sits beside wave 0 and stays idle here: the pivot chain is issue-sensitive)
        const int worker = (wave & 3) ? (wave - 1 - (wave >> 2)) : -1;      // 1,2,3,5,6,7 -> 0..5
        if (worker >= 0 && s > 0) {
            const int po = (s - 1) * 16;
            const int w = 8 - s;                        // tile rows/cols s..7
            const int ntile = w * (w + 1) / 2;
            for (int t = 1 + worker; t < ntile; t += 6) {   // t = 0 is tile (s,s): done by wave 0
                int i = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                while (i * (i + 1) / 2 > t) --i;
                const int j = t - i * (i + 1) / 2;
                const int ti = s + i, tj = s + j;
                double* Ct = As + (ti * 16 + 4 * lb + lq) * CGP_LDA + tj * 16 + lu;
                double acc[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) acc[J] = Ct[4 * J];
                tile_mma_nt<true>(acc, As + (ti * 16) * CGP_LDA + po, As + (tj * 16) * CGP_LDA + po, lr, lu, lq);
#pragma unroll
                for (int J = 0; J < 4; ++J) Ct[4 * J] = acc[J];
            }
        }
        if (wave == 0) {
            // lanes 0..15: rows of T; lanes 16..31: rows of the identity (-> T^-T); others idle.
            // Column j: pivot by readlane, 1/sqrt by rsq + one-shot refinement, column j+1 is updated at
            // once through readlane (it is on the critical chain), columns >= j+2 one step later
            // from an LDS broadcast of the scaled column (fills the latency of the next chain).
            double v[16], lprev = 0., pv;
            d2 bc[8];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double t = 0.;
                if (lane < 16) t = As[(o + lane) * CGP_LDA + o + c];
                else if (lane < 32) t = (c == lane - 16) ? 1. : 0.;
                v[c] = t;
            }
            pv = v[0];
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double d = readlane_f64(pv, j);
                if (!(d > 0.) && bad == 0) bad = o + j + 1;
                double l, sd;
                pivot_scale(d, v[j], l, sd);             // lane j's own l is never used as a multiplier
                v[j] = (lane == j) ? sd : l;
                if (j > 0) {       // delayed updates of column j-1 on columns >= j+1
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) {
                        const double lc = (c & 1) ? bc[c >> 1].y : bc[c >> 1].x;
                        v[c] = __builtin_fma(-lprev, lc, v[c]);
                    }
                }
                if (j < 15) {
                    // next pivot straight from the pivot lane's own v - l^2 (no readlane on the chain)
                    pv = __builtin_fma(-l, l, v[j + 1]);
                    const double lc = readlane_f64(l, j + 1);
                    v[j + 1] = __builtin_fma(-l, lc, v[j + 1]);
                }
                if (j < 14) {
                    Lc[(j & 1) * 64 + lane] = l;          // every lane writes (no exec-mask branch)
                    wave_lds_fence();
#pragma unroll
                    for (int c2 = (j + 2) >> 1; c2 < 8; ++c2) bc[c2] = *(const d2*)(Lc + (j & 1) * 64 + 2 * c2);
                }
                lprev = l;
            }
            __builtin_amdgcn_s_setprio(0);
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) As[(o + lane) * CGP_LDA + o + c] = (c <= lane) ? v[c] : 0.;
            } else if (lane < 32) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Ws[s * 256 + (lane - 16) * 16 + c] = (c >= lane - 16) ? v[c] : 0.;
            }
        }
        __syncthreads();
        CGP_STAMP(3 + 4 * s);
        // I2: sub-panel  P[ti] = A[ti][s] * T^-T  (B[k][n] = Ws[s][k][n]); wave 0 takes ti = s+1 and then
        // the diagonal tile (s+1,s+1), the other waves share ti >= s+2
        for (int ti = (wave == 0 ? s + 1 : s + 1 + wave); ti < (wave == 0 ? min(s + 2, 8) : 8);
             ti += CGP_K3_WAVES - 1) {
            double acc[4] = {0., 0., 0., 0.};
            const double* Arow = As + (ti * 16) * CGP_LDA + o;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                d2 a = *(const d2*)(Arow + lr * CGP_LDA + 8 * t + 2 * lq);
                double b0[4], b1[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    b0[J] = Ws[s * 256 + (8 * t + 2 * lq) * 16 + 4 * J + lu];
                    b1[J] = Ws[s * 256 + (8 * t + 2 * lq + 1) * 16 + 4 * J + lu];
                }
#pragma unroll
                for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.x, b0[J], acc[J]);
#pragma unroll
                for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.y, b1[J], acc[J]);
            }
            // acc[J] at lane (u, blk, q) = P[16 ti + 4 blk + q][4 J + u]
            wave_lds_fence();
#pragma unroll
            for (int J = 0; J < 4; ++J) As[(ti * 16 + 4 * lb + lq) * CGP_LDA + o + 4 * J + lu] = acc[J];
        }
        if (wave == 0 && s < 7) {
            // (e0): tile (s+1,s+1) -= P[s+1] P[s+1]^T, P[s+1] just written by this wave
            wave_lds_fence();
            const int t1 = s + 1;
            double* Ct = As + (t1 * 16 + 4 * lb + lq) * CGP_LDA + t1 * 16 + lu;
            double acc[4];
#pragma unroll
            for (int J = 0; J < 4; ++J) acc[J] = Ct[4 * J];
            tile_mma_nt<true>(acc, As + (t1 * 16) * CGP_LDA + o, As + (t1 * 16) * CGP_LDA + o, lr, lu, lq);
#pragma unroll
            for (int J = 0; J < 4; ++J) Ct[4 * J] = acc[J];
        }
        __syncthreads();
        CGP_STAMP(4 + 4 * s);
    }

    return bad;
}

template <int GEN>
__device__ __forceinline__ void potrf_diag_body(
    double* __restrict__ M, long ld, long sM, int c0, double* __restrict__ wsW, long sW,
    double* __restrict__ logdet_parts, long sLd, int* __restrict__ info, int blk, const GenSrc& g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                          // [128][CGP_LDA]
    double* Ws = smem + CGP_NB * CGP_LDA;       // [8][16][16]  T^-T blocks (upper triangular)
    double* Lc = Ws + 8 * 256;                  // [2][64] column broadcast buffer of phase (a)
    const int b = blockIdx.z;
    M += (long)b * sM + (long)c0 * ld + c0;
    wsW += (long)b * sW + (long)blk * (8 * 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    CGP_STAMP(0);
    if (GEN != 0) {
        // first touch of this block: its entries are computed (see GenSrc), sub-tiles on and below the block diagonal
        constexpr int GD = GEN == 2 ? 2 : 1;
        __shared__ __attribute__((aligned(16))) double etab[128];
        rbf_stage_table(etab);
        const GenObj o = gen_obj<GD>(g, b);
        // a thread's column pair is the same in every pass and the row is the same for the whole wave: the column points are
        // loaded once, the 128 row descriptors go through LDS (Ws is free until the factorisation starts): no global load
        // inside the loop
        GenRow* rws = (GenRow*)Ws;                           // 128 x 32 B = 4 KB of the 16 KB
        if (tid < CGP_NB) rws[tid] = gen_row<GD>(g, o, g.row0 + tid);
        const int c = (tid & 63) * 2, cc = g.col0 + c;
        double ux0, uy0 = 0., ux1, uy1 = 0.;
        if (GD == 1) { const d2 t = *(const d2*)(o.x + cc); ux0 = t.x; ux1 = t.y; }
        else { const d2 t0 = *(const d2*)(o.x + 2 * (long)cc), t1 = *(const d2*)(o.x + 2 * (long)cc + 2); ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y; }
        __syncthreads();
        constexpr int NCH = CGP_NB * (CGP_NB / 2) / CGP_K3_THREADS;
#pragma unroll 4
        for (int k = 0; k < NCH; ++k) {
            // (computed for every lane, stored for the sub-tiles on and below the diagonal: no branch around the arithmetic, so
            // the unrolled passes interleave -- the dependent fp64 chain of one entry is ~25 operations of 30-40 cycles each)
            const int r = (tid >> 6) + k * (CGP_K3_THREADS / 64);
            const int i = g.row0 + r;
            const GenRow rw = rws[r];
            d2 v;
            v.x = gen_val<GD>(o, rw, i, cc, ux0, uy0, etab);
            v.y = gen_val<GD>(o, rw, i, cc + 1, ux1, uy1, etab);
            if ((c >> 4) <= (r >> 4)) *(d2*)(As + r * CGP_LDA + c) = v;
        }
        __syncthreads();
    } else {
    // load the block by direct global -> LDS loads: one wave instruction = one full row (128 doubles) into
    // As[r][0..127]; the part right of the diagonal tile is never read
    for (int r = wave; r < CGP_NB; r += CGP_K3_THREADS / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(M + (long)r * ld + 2 * lane),
                                         (__attribute__((address_space(3))) void*)(As + r * CGP_LDA), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }
    CGP_STAMP(1);

    const int bad = potrf128_lds(As, Ws, Lc);

    CGP_STAMP(40);
    // write L back (tiles on and below the block diagonal), T^-T blocks, logdet, info
    {
        constexpr int NCH = CGP_NB * (CGP_NB / 2) / CGP_K3_THREADS;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = tid + k * CGP_K3_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            if ((c >> 4) <= (r >> 4)) *(d2*)(M + (long)r * ld + c) = *(const d2*)(As + r * CGP_LDA + c);
        }
    }
    for (int idx = tid; idx < 8 * 256; idx += CGP_K3_THREADS) wsW[idx] = Ws[idx];
    double lg = 0.;
    if (tid < CGP_NB) lg = 2. * log(As[tid * CGP_LDA + tid]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    double* red = Lc + 128;                     // 2 doubles after the broadcast buffer
    if (lane == 0 && wave < 2) red[wave] = lg;
    __syncthreads();
    if (tid == 0) {
        logdet_parts[(long)b * sLd + blk] = red[0] + red[1];
        if (bad != 0 && info[b] == 0) info[b] = c0 + bad;
    }
    CGP_STAMP(41);
}

__global__ __launch_bounds__(CGP_K3_THREADS) void k_potrf_diag(
    double* __restrict__ M, long ld, long sM, int c0, double* __restrict__ wsW, long sW,
    double* __restrict__ logdet_parts, long sLd, int* __restrict__ info, int blk) {
    potrf_diag_body<0>(M, ld, sM, c0, wsW, sW, logdet_parts, sLd, info, blk, GenSrc());
}
// the same with the block's entries computed instead of loaded (first touch, see GenSrc); GEN = 1 / 2
template <int GEN>
__global__ __launch_bounds__(CGP_K3_THREADS) void k_potrf_diag_gen(
    double* __restrict__ M, long ld, long sM, int c0, double* __restrict__ wsW, long sW,
    double* __restrict__ logdet_parts, long sLd, int* __restrict__ info, int blk, GenSrc g) {
    potrf_diag_body<GEN>(M, ld, sM, c0, wsW, sW, logdet_parts, sLd, info, blk, g);
}

// ------------------------------------------------------------------------------------
// K4: panel solve  X = A L_kk^-T  for the rows below a diagonal block (right-looking over the
// eight 16-column tiles): X_j = S_j T_j^-T, then S_i -= X_j L_ij^T for i > j.
// One wave owns 16 panel rows; S (8 tiles) lives in registers in the 4-block MFMA result
// layout; a finished X_j goes through a wave-private LDS tile to be re-read as an A operand.
// L_kk (strictly lower 16x16 tiles) and the T_j^-T blocks are staged in LDS once per workgroup.
// ------------------------------------------------------------------------------------

template <int K4W, int GEN = 0>
__global__ __launch_bounds__(64 * K4W) void k_trsm_panel(
    double* __restrict__ M, long ld, long sM, int c0, int r0, const double* __restrict__ wsW,
    long sW, int blk, int nrows, GenSrc g = GenSrc(), int xcdB = 0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ls = smem;                               // [128][CGP_LDA] strictly-lower tiles used
    double* Ws = smem + CGP_NB * CGP_LDA;            // [8][16][16]
    int b = blockIdx.z, bx = blockIdx.x;
    if (xcdB > 0 && !xcd_affine(b, bx, xcdB)) return;
    M += (long)b * sM;
    wsW += (long)b * sW + (long)blk * (8 * 256);
    const double* Lg = M + (long)c0 * ld + c0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    // wave-private 16 x 16 tile, kept in the part of Ls that is never loaded (tiles on/above the diagonal):
    // tile rows g = 0, 1 offer columns 16..127 (6 slots of 18), g = 2 columns 32..127 (5 slots); row stride CGP_LDA
    const int xg = wave < 6 ? 0 : (wave < 12 ? 1 : 2), xt = wave - 6 * xg;
    double* Xs = Ls + xg * 16 * CGP_LDA + (xg == 2 ? 32 : 16) + 18 * xt;
    double* Ap = M + (long)(r0 + bx * (16 * K4W) + wave * 16 + 4 * lb + lq) * ld + c0 + 4 * lu;
    // nrows need not fill the last workgroup (the tall rows of small objects come in units of 64): its surplus waves help
    // to stage L_kk and then have no rows of their own
    const bool has_rows = (int)(bx * (16 * K4W) + wave * 16) < nrows;

    CGP_STAMP(50);
    // This wave's 16 x 128 panel rows in MFMA result layout.  The column slot (J, u) of a 16-wide
    // tile stands for column 4 u + J (any fixed permutation is legal as long as the B operands use
    // the same one), so a lane's four J values are 4 contiguous doubles: 16-byte loads/stores.
    double S[8][4];
    if (GEN == 0 && has_rows) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            d2 t0 = *(const d2*)(Ap + jb * 16), t1 = *(const d2*)(Ap + jb * 16 + 2);
            S[jb][0] = t0.x; S[jb][1] = t0.y; S[jb][2] = t1.x; S[jb][3] = t1.y;
        }
    }
    // L_kk -> LDS by direct global -> LDS loads (global_load_lds_dwordx4): one wave instruction moves one full
    // row of the block (64 lanes x 16 B = 128 doubles) to Ls[r][0..127], all rows of a wave in flight at once.
    // Only the strictly-lower tiles are read later; rows 0..15 have none (and host the wave-private X tiles).
    constexpr int K4T = 64 * K4W;
    for (int r = 16 + wave; r < CGP_NB; r += K4W)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Lg + (long)r * ld + 2 * lane),
                                         (__attribute__((address_space(3))) void*)(Ls + r * CGP_LDA), 16, 0, 0);
    {
        d2 w[1024 / K4T];
#pragma unroll
        for (int k = 0; k < 1024 / K4T; ++k) w[k] = *(const d2*)(wsW + 2 * (tid + k * K4T));
#pragma unroll
        for (int k = 0; k < 1024 / K4T; ++k) *(d2*)(Ws + 2 * (tid + k * K4T)) = w[k];
    }
    if (GEN != 0) {
        // first touch of these rows (panel 0): computed, not loaded (see GenSrc), while the loads of L_kk above are in flight
        constexpr int GD = GEN == 2 ? 2 : 1;
        __shared__ __attribute__((aligned(16))) double etab[128];
        rbf_stage_table(etab);
        __syncthreads();
        if (has_rows) {
            const GenObj o = gen_obj<GD>(g, b);
            const int i = r0 + bx * (16 * K4W) + wave * 16 + 4 * lb + lq;
            const GenRow rw = gen_row<GD>(g, o, i);
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                const int cc = c0 + jb * 16 + 4 * lu;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    double ux0, uy0 = 0., ux1, uy1 = 0.;
                    if (GD == 1) { const d2 t = *(const d2*)(o.x + cc + e); ux0 = t.x; ux1 = t.y; }
                    else { const d2 t0 = *(const d2*)(o.x + 2 * (long)(cc + e)), t1 = *(const d2*)(o.x + 2 * (long)(cc + e) + 2); ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y; }
                    S[jb][e] = gen_val<GD>(o, rw, i, cc + e, ux0, uy0, etab);
                    S[jb][e + 1] = gen_val<GD>(o, rw, i, cc + e + 1, ux1, uy1, etab);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CGP_STAMP(51);
    if (!has_rows) return;

#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        CGP_STAMP(52 + jb);
        wave_lds_fence();
        // S_jb -> LDS (result layout) -> A operand;  X_jb = S_jb * T_jb^-T
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu) = (d2){S[jb][0], S[jb][1]};
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu + 2) = (d2){S[jb][2], S[jb][3]};
        wave_lds_fence();
        double x[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d2 a = *(const d2*)(Xs + lr * CGP_LDA + 8 * t + 2 * lq);
            const double* w0 = Ws + jb * 256 + (8 * t + 2 * lq) * 16 + 4 * lu;     // B[k][4u + J]
            d2 p0 = *(const d2*)(w0), p1 = *(const d2*)(w0 + 2), p2 = *(const d2*)(w0 + 16), p3 = *(const d2*)(w0 + 18);
            x[0] = mfma4(a.x, p0.x, x[0]); x[1] = mfma4(a.x, p0.y, x[1]);
            x[2] = mfma4(a.x, p1.x, x[2]); x[3] = mfma4(a.x, p1.y, x[3]);
            x[0] = mfma4(a.y, p2.x, x[0]); x[1] = mfma4(a.y, p2.y, x[1]);
            x[2] = mfma4(a.y, p3.x, x[2]); x[3] = mfma4(a.y, p3.y, x[3]);
        }
        *(d2*)(Ap + jb * 16) = (d2){x[0], x[1]};
        *(d2*)(Ap + jb * 16 + 2) = (d2){x[2], x[3]};
        if (jb < 7) {
            wave_lds_fence();
            *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu) = (d2){x[0], x[1]};
            *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu + 2) = (d2){x[2], x[3]};
            wave_lds_fence();
            // S_i -= X_jb * L[i][jb]^T  for every later tile i  (B row of slot (J,u) = 16 i + 4 u + J)
            d2 a[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = -*(const d2*)(Xs + lr * CGP_LDA + 8 * t + 2 * lq);
#pragma unroll
            for (int i = jb + 1; i < 8; ++i) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    d2 bb[4];
#pragma unroll
                    for (int J = 0; J < 4; ++J)
                        bb[J] = *(const d2*)(Ls + (i * 16 + 4 * lu + J) * CGP_LDA + jb * 16 + 8 * t + 2 * lq);
#pragma unroll
                    for (int J = 0; J < 4; ++J) S[i][J] = mfma4(a[t].x, bb[J].x, S[i][J]);
#pragma unroll
                    for (int J = 0; J < 4; ++J) S[i][J] = mfma4(a[t].y, bb[J].y, S[i][J]);
                }
            }
        }
    }
    CGP_STAMP(60);
}

// ------------------------------------------------------------------------------------
// K5: C = beta C + alpha A B^T on v_mfma_f64_4x4x4_4b_f64 -- the trailing SYRK/GEMM update
// (alpha = -1, beta = 1), the predictive covariance K** - W W^T and the explicit inverse
// W W^T.  A: rows x K, B: rows x K, both row-major with the same ld and K contiguous.
//   workgroup = WM x WN waves, wave tile (TM/WM) x (TN/WN); KB = 16 deep stages double-buffered
//   in LDS (global -> registers -> LDS, one barrier per stage).
//   Register tiling for the 4-block instruction: block blk of a wave owns the rows
//   16 r + 4 blk + (0..3) (r = 0..MT-1: the A fragment is fetched exactly like a 16-row MFMA
//   operand) and ALL column groups J of the wave tile: the B fragment of group J
//   (cols 4 J + (l & 3)) is the same in the four blocks, i.e. a broadcast LDS read.
//   acc[r][J] at lane l = C[16 r + 4 blk + (l >> 4)][4 J + (l & 3)].
//   Each lane reads 2 consecutive k per ds_read_b128 and uses them in two MFMAs (the k order
//   inside a stage is permuted identically for A and B, which leaves the sum unchanged).
//   tri = 1: only tiles on/below the diagonal of the (square-tiled) C region are launched:
//   tile rows [0, nc) form the triangle, rows [nc, nr) are full.
// ------------------------------------------------------------------------------------
#ifndef CGP_EARLYC_MAX
#define CGP_EARLYC_MAX 8
#endif
template <int TM, int TN, int WM, int WN> struct MT_OK_FOR_GEN { static constexpr bool value = (TM / WM / 16) * (TN / WN / 4) <= CGP_EARLYC_MAX; };
// KB = stage depth (16 or 32); LDS row stride KB + 2 doubles keeps every row 16-B aligned
// ABLATE (diagnostic builds only, tools/gemm_ablate.hip): 1 = no global loads inside the loop,
// 2 = additionally no LDS fragment reads inside the loop (wrong results, timing only)
#ifndef CGP_SMALL_WPE
#define CGP_SMALL_WPE 1      // minimum waves per SIMD asked of the compiler for the 64 x 64 form (1 = no constraint; scan: 4.3)
#endif
template <int TM, int TN, int WM, int WN, int KB, int ABLATE = 0, int GEN = 0>
__global__ __launch_bounds__(64 * WM * WN, (TM == 64 && TN == 64) ? CGP_SMALL_WPE : 1) void k_gemm_nt(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int tile_skip, int ktri,
    int tile_end = 0, const int2* __restrict__ tab = nullptr, int ntab = 0, GenSrc g = GenSrc(), int xcdB = 0,
    int rows_valid = 0) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int MT = TM / WM / 16;          // A fragments (16 rows each) per wave
    constexpr int NJ = TN / WN / 4;           // B column groups (4 cols each) per wave
    constexpr int KS = KB + 2;
    constexpr int CPR = KB / 2;               // 16-byte chunks per row of a stage
    constexpr int PA = TM * CPR / NTH, PB = TN * CPR / NTH;   // chunks per thread per stage
    constexpr int RSTEP = NTH / CPR;
    static_assert(PA >= 1 && PB >= 1, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(16))) double As[2][TM * KS];
    __shared__ __attribute__((aligned(16))) double Bs[2][TN * KS];
    int bz = blockIdx.z, bx0 = blockIdx.x;
    if (xcdB > 0 && !xcd_affine(bz, bx0, xcdB)) return;        // (see xcd_affine: all tiles of an object on one XCD)
    C += (long)bz * sC;
    A += (long)bz * sAB;
    B += (long)bz * sAB;
    // GEN (small wave tiles only): the update is the first touch of its C region -- the "beta C" term is computed (GenSrc)
    constexpr int GD = GEN == 2 ? 2 : 1;
    static_assert(GEN == 0 || MT_OK_FOR_GEN<TM, TN, WM, WN>::value, "GEN needs the early-C form");
    __shared__ __attribute__((aligned(16))) double etab[GEN != 0 ? 128 : 2];
    if (GEN != 0) {
        rbf_stage_table(etab);
        __syncthreads();
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    const int srow = tid / CPR, sch = (tid % CPR) * 2;   // staging: RSTEP rows x CPR chunks of 16 B
    // Triangular tile map for TM = RT * TN (RT = 1 or 2): tile row ti of the triangle (ti < nt = nc / RT)
    // holds the RT (ti + 1) column tiles that touch the lower triangle.
    constexpr int RT = TM / TN;
    static_assert(TM == RT * TN && (RT == 1 || RT == 2), "tile shape");
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot_all = ntri + (nr - (tri ? nt : 0)) * nc;
    const int ntot = tile_end > 0 ? min(tile_end, ntot_all) : ntot_all;     // tile_end: this launch covers tiles [tile_skip, tile_end)
    // Persistent tile loop: a launch with fewer workgroups than tiles walks the tile list with stride
    // gridDim.x (bulk updates are launched with <= 240 workgroups of this 1-per-CU kernel so that some
    // CUs stay free for the latency-critical leaf kernels of the other stream).
    // tab != nullptr: the launch grid walks a host-built tile list instead (XCD-aligned order, see launch_gemm);
    // entries with x < 0 are padding
    for (int bid = bx0 + tile_skip; bid < (tab ? ntab : ntot); bid += gridDim.x) {
    int ti, tj;
    if (tab) {
        const int2 t = tab[bid];
        if (t.x < 0) continue;
        ti = t.x;
        tj = t.y;
    } else {
        // Block ids are dealt round-robin over the 8 XCDs; with the row-major tile order the tiles that
        // run together on one XCD span ~16 tile rows x 8 columns, i.e. 24 shared panels per 128 tiles.
        // (A contiguous chunk per XCD was measured WORSE: +15 % HBM fetch, same time.)
        if (bid < ntri) {
            int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
            while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
            while (RT * i * (i + 1) / 2 > bid) --i;
            ti = i;
            tj = bid - RT * i * (i + 1) / 2;
        } else {
            int r = bid - ntri;
#ifdef CGP_GEMM_XCD2D
            // experiment (tools/gemm_ablate.hip): 2 x 4 grid of XCDs over the rectangular part, XCD (xr, xc)
            // owns tile rows = xr (mod 2) and tile columns = xc (mod 4)
            {
                const int nrr = nr - (tri ? nt : 0);
                if ((nrr & 1) == 0 && (nc & 3) == 0) {
                    const int x = r & 7, loc = r >> 3, xr = x >> 2, xc = x & 3, ncx = nc >> 2;
                    const int lr_ = loc / ncx, lc_ = loc % ncx;
                    r = (lr_ * 2 + xr) * nc + lc_ * 4 + xc;
                }
            }
#endif
            ti = (tri ? nt : 0) + r / nc;
            tj = r % nc;
        }
    }
    // ktri: A and B are rows of an upper-triangular matrix (row t is zero left of column t, W = L^-T of
    // the explicit inverse): the products of tile row ti vanish for k < ti * TM, the K loop starts there
    const int kstart = ktri ? ti * TM : 0;
    const double* Ag = A + (long)(ti * TM) * ld + kstart;
    const double* Bg = B + (long)(tj * TN) * ld + kstart;
    // Waves without useful work sit the tile out (round 4; they still stage and keep the barriers): (i) rows_valid > 0: rows of
    // the C region from rows_valid on carry nothing (the tall rows of a small object are padded to the 64-row tile, 20 of 64
    // are real: a quarter of the tiles of a 512-point light curve are tall-row tiles) -- a wave whose rows all lie beyond
    // neither computes nor stores, and its share of the A stage is not even loaded; (ii) a tile on the diagonal of a
    // triangular region: the wave tiles entirely above the diagonal.  What IS computed is computed as before: same bits.
    const bool wave_on = !((rows_valid > 0 && ti * TM + wm * (TM / WM) >= rows_valid) ||
                           (tri && RT == 1 && bid < ntri && ti == tj && wn * (TN / WN) > wm * (TM / WM) + (TM / WM) - 1));
    const bool arow_on = !(rows_valid > 0 && ti * TM + srow >= rows_valid);        // (PA = 1 for the small form: one row per thread)

    double acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.;
    // acc[i][j] at lane l = C[16 i + 4 blk + (l >> 4)][4 j + (l & 3)],  blk = (l >> 2) & 3
    double* Cg = C + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN + wn * (TN / WN) + lp;
    // Small wave tiles (the thin updates): the C values are fetched BEFORE the K loop, so the read-modify-write
    // latency of the epilogue hides under it (8 more registers; the large form keeps its 4 waves per SIMD instead)
    constexpr bool EARLYC = (MT * NJ <= CGP_EARLYC_MAX);
    double cv[EARLYC ? MT : 1][EARLYC ? NJ : 1];
    if (GEN != 0) {
        // (computed below, once the first two stage loads are on their way)
    } else if (EARLYC && beta != 0. && wave_on) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) cv[EARLYC ? i : 0][EARLYC ? j : 0] = Cg[(long)(i * 16) * ldc + j * 4];
    }

    // Global -> registers two stages ahead, registers -> LDS one stage ahead: a load has two
    // full stages of MFMA work to land (thin K = 128 updates are otherwise latency-bound).
    d2 ra[2][PA], rb[2][PB];
    const int nk = (K - kstart) / KB;
    auto gload = [&](auto par, int kt) {
        constexpr int P = decltype(par)::value;
        const int k1 = kt * KB + sch;
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[P][p] = (PA > 1 || arow_on) ? *(const d2*)(Ag + (long)(srow + RSTEP * p) * ld + k1) : (d2){0., 0.};
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[P][p] = *(const d2*)(Bg + (long)(srow + RSTEP * p) * ld + k1);
    };
    auto lstore = [&](auto par, int buf) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int p = 0; p < PA; ++p) *(d2*)(&As[buf][(srow + RSTEP * p) * KS + sch]) = ra[P][p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *(d2*)(&Bs[buf][(srow + RSTEP * p) * KS + sch]) = rb[P][p];
    };
    auto compute = [&](int cur, int kt) {
        if (!wave_on) return;
        d2 fa[MT], fb[NJ];
        if (ABLATE >= 2) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = ra[0][0] + (double)(i + kt);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = rb[0][0] + (double)(j - kt);
        }
#pragma unroll
        for (int t = 0; t < KB / 8; ++t) {
            if (ABLATE < 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    fa[i] = *(const d2*)(&As[cur][(wm * (TM / WM) + i * 16 + lr) * KS + 8 * t + 2 * lq]);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    fb[j] = *(const d2*)(&Bs[cur][(wn * (TN / WN) + j * 4 + lp) * KS + 8 * t + 2 * lq]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[i].x, fb[j].x, acc[i][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[i].y, fb[j].y, acc[i][j]);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // nk is even (K is a multiple of 32).  The prefetches are UNCONDITIONAL (the stage index is clamped at
    // the end of the K loop, the last two are redundant re-loads): with a conditional load the compiler
    // cannot count the loads in flight and waits for vmcnt(0) before every LDS store, which turns the
    // two-stages-ahead prefetch into a full memory round trip per stage.
    gload(P0{}, 0);
    gload(P1{}, 1);
    __builtin_amdgcn_sched_barrier(0);      // keep the loads where they are written (see above)
    if (GEN != 0 && wave_on) {
        const GenObj o = gen_obj<GD>(g, bz);
        const int rr0 = g.row0 + ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq, cc0 = g.col0 + tj * TN + wn * (TN / WN) + lp;
        double ux[NJ], uy[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ux[j] = o.x[(long)(cc0 + 4 * j) * GD];
            uy[j] = GD == 2 ? o.x[(long)(cc0 + 4 * j) * GD + 1] : 0.;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const GenRow rw = gen_row<GD>(g, o, rr0 + 16 * i);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                cv[EARLYC ? i : 0][EARLYC ? j : 0] = gen_val<GD>(o, rw, rr0 + 16 * i, cc0 + 4 * j, ux[j], uy[j], etab);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    lstore(P0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even stage kt: registers P1 hold stage kt+1, P0 is free
        if (ABLATE == 0) gload(P0{}, min(kt + 2, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        compute(0, kt);
        lstore(P1{}, 1);
        __syncthreads();
        // odd stage kt+1: registers P0 hold stage kt+2, P1 is free
        if (ABLATE == 0) gload(P1{}, min(kt + 3, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        compute(1, kt + 1);
        lstore(P0{}, 0);                         // (redundant after the last stage; unconditional on purpose)
        __syncthreads();
    }

    if (!wave_on) {
        // (nothing to store)
    } else if (EARLYC && beta != 0.) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j] + beta * cv[EARLYC ? i : 0][EARLYC ? j : 0];
    } else if (beta != 0.) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double* p = Cg + (long)(i * 16) * ldc + j * 4;
                *p = alpha * acc[i][j] + beta * (*p);
            }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j];
    }
    }   // persistent tile loop
}

// ------------------------------------------------------------------------------------
// K7: finish.  Block t < m: mu_t = W_t . z + mean_t, var_t = (amp_train + kss_noise) - |W_t|^2.
// Block m: loglik = -1/2 z.z - n/2 log(2 pi) - 1/2 logdet  (Gaussian_process.py:68-73).
// z is the solved residual row (row zrow of the tall part).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finish(
    const double* __restrict__ M, long ld, long sM, int Np, int wrow0, int zrow, int m,
    const double* __restrict__ mean_new, const KParams* __restrict__ kpp,
    const double* __restrict__ logdet_parts, long sLd, int nblk, const int* __restrict__ ns,
    double* __restrict__ mu, double* __restrict__ var, long sOut, double* __restrict__ scal, int kps = 0) {
    const int b = blockIdx.z, t = blockIdx.x;
    kpp += (long)b * kps;
    M += (long)b * sM;
    const double* z = M + (long)zrow * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double red[2][4];
    double s1 = 0., s2 = 0.;
    if (t < m) {
        const double* w = M + (long)(wrow0 + t) * ld;
        for (int c = tid * 2; c < Np; c += 512) {
            d2 wv = *(const d2*)(w + c), zv = *(const d2*)(z + c);
            s1 += wv.x * zv.x + wv.y * zv.y;
            s2 += wv.x * wv.x + wv.y * wv.y;
        }
    } else {
        for (int c = tid * 2; c < Np; c += 512) {
            d2 zv = *(const d2*)(z + c);
            s1 += zv.x * zv.x + zv.y * zv.y;
        }
        if (tid < nblk) s2 = logdet_parts[(long)b * sLd + tid];
        for (int k = tid + 256; k < nblk; k += 256) s2 += logdet_parts[(long)b * sLd + k];
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
            mu[(long)b * sOut + t] = s1 + (mean_new ? mean_new[t] : 0.);
            var[(long)b * sOut + t] = (kpp->amp_train + kpp->kss_noise) - s2;
        } else {
            const double n = (double)ns[b];
            scal[b * 2 + 0] = -0.5 * s1 - 0.5 * n * 1.8378770664093453 - 0.5 * s2;   // log(2 pi)
            scal[b * 2 + 1] = s2;
        }
    }
}

// K7p: finish of PACKED small objects (GenSrc: `pack` objects of <= hs points per 128 x 128 tile).  Block t < pack m: local
// object lo = t / m, its cross row t of the tile's tall rows against ITS solved residual row zrow + lo -- both are zero
// outside the object's columns, so the 128-column dot products are the object's own, summed in k_finish's order.  Blocks
// pack m + lo: the object's log-likelihood, its log-determinant from the diagonal of ITS block of L (the 64-lane tree of
// k_potrf_diag's reduction with zeros beyond the block: the bits of the object factored alone in a tile).
__global__ __launch_bounds__(256) void k_finish_packed(
    const double* __restrict__ M, long ld, long sM, int Np, int zrow, int m, int pack, int hs, int nobj,
    const double* __restrict__ mean_new, const KParams* __restrict__ kpp, const int* __restrict__ ns,
    double* __restrict__ mu, double* __restrict__ var, long sOut, double* __restrict__ scal) {
    const int b = blockIdx.z, t = blockIdx.x;
    const int lo = t < pack * m ? t / m : t - pack * m;
    const int obj = b * pack + lo;
    if (obj >= nobj) return;
    M += (long)b * sM;
    const double* z = M + (long)(zrow + lo) * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double red[2][4];
    double s1 = 0., s2 = 0.;
    if (t < pack * m) {
        const double* w = M + (long)(Np + t) * ld;
        for (int c = tid * 2; c < Np; c += 512) {
            d2 wv = *(const d2*)(w + c), zv = *(const d2*)(z + c);
            s1 += wv.x * zv.x + wv.y * zv.y;
            s2 += wv.x * wv.x + wv.y * wv.y;
        }
    } else {
        for (int c = tid * 2; c < Np; c += 512) {
            d2 zv = *(const d2*)(z + c);
            s1 += zv.x * zv.x + zv.y * zv.y;
        }
        // 2 log L_ii of this object's block in lanes 0 .. hs - 1 of wave 0 (k_potrf_diag: one value per thread, zeros = log 1 beyond)
        if (tid < hs) s2 = 2. * log(M[(long)(lo * hs + tid) * ld + lo * hs + tid]);
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
        if (t < pack * m) {
            s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            const int tt = t - lo * m;
            mu[(long)obj * sOut + tt] = s1 + (mean_new ? mean_new[tt] : 0.);
            var[(long)obj * sOut + tt] = (kpp->amp_train + kpp->kss_noise) - s2;
        } else {
            s2 = red[1][0] + red[1][1];          // (k_potrf_diag adds the sums of its waves 0 and 1)
            const double n = (double)ns[obj];
            scal[obj * 2 + 0] = -0.5 * s1 - 0.5 * n * 1.8378770664093453 - 0.5 * s2;   // log(2 pi)
            scal[obj * 2 + 1] = s2;
        }
    }
}

// resident unpacked batch ([B][128] per quantity) -> packed arrays ([B][hs], i.e. [tiles][128]); objects beyond B: zeros, n = 0
__global__ void k_pack_small(const double* __restrict__ x, const double* __restrict__ r, const double* __restrict__ d,
                             const int* __restrict__ ns, double* __restrict__ px, double* __restrict__ pr,
                             double* __restrict__ pd, int* __restrict__ pns, int B, int Bpad, int hs, int dim) {
    const int obj = blockIdx.x, tid = threadIdx.x;
    if (obj >= Bpad) return;
    const bool real = obj < B;
    if (tid == 0) pns[obj] = real ? ns[obj] : 0;
    for (int i = tid; i < hs; i += blockDim.x) {
        pr[(long)obj * hs + i] = real ? r[(long)obj * 128 + i] : 0.;
        pd[(long)obj * hs + i] = real ? d[(long)obj * 128 + i] : 0.;
        for (int k = 0; k < dim; ++k) px[((long)obj * hs + i) * dim + k] = real ? x[((long)obj * 128 + i) * dim + k] : 0.;
    }
}

// ------------------------------------------------------------------------------------
// K8: moment sums for the analytic gradient of the log marginal likelihood
//   dlogL/dtheta = 1/2 sum_ij M_ij dK_ij/dtheta,   M = alpha alpha^T - K^-1,  alpha = K^-1 r.
// With K_ij = amp_train E_ij + delta_ij (diag_i + nug2), E = exp(-q/2), every derivative of the RBF
// families of kernel.py is a combination of five sums over i, j < n (the host applies the chain rule
// from the kp parametrisation to the caller's hyperparameters):
//   S[0] = sum M_ij E_ij      S[1] = sum M_ij E_ij d0^2     S[2] = sum M_ij E_ij d0 d1
//   S[3] = sum M_ij E_ij d1^2     S[4] = sum_i M_ii
// Kinv is read from its lower triangle (off-diagonal terms count twice).  64 x 128 tile per
// workgroup like K1; E is recomputed from the coordinates (no second N x N read).  Partial sums per
// workgroup go to part[block][8]; k_grad_final adds them in a fixed order (bit-reproducible).
// ------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_grad_sums(
    const double* __restrict__ Kinv, long ld, const double* __restrict__ alpha, const double* __restrict__ x,
    const int* __restrict__ ns, const KParams* __restrict__ kpp, double* __restrict__ part,
    long sK = 0, long sAlpha = 0, long sx = 0, int kps = 0) {
    const int bz = blockIdx.z;                 // object of a batch (strides 0 / grid.z = 1 for a single object)
    Kinv += (long)bz * sK;
    alpha += (long)bz * sAlpha;
    x += (long)bz * sx;
    kpp += (long)bz * kps;
    part += (long)bz * gridDim.x * gridDim.y * 8;
    const int col0 = blockIdx.x * 128, row0 = blockIdx.y * 64;
    const int bl = blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[5] = {0., 0., 0., 0., 0.};
    __shared__ double rx[64 * 2], ra[64], red[4][5];
    __shared__ __attribute__((aligned(16))) double etab[128];
    const int n = ns[bz];
    rbf_stage_table(etab);
    if (col0 <= row0 + 63 && row0 < n) {
        const KParams kp = *kpp;
        const double yinv = rbf_arg_const<DIM>(kp);
        if (tid < 64 * DIM) rx[tid] = x[(long)row0 * DIM + tid];
        if (tid < 64) ra[tid] = alpha[row0 + tid];
        const int c = col0 + 2 * lane;
        double ux0, uy0 = 0., ux1, uy1 = 0.;
        if (DIM == 1) {
            d2 t = *(const d2*)(x + c);
            ux0 = t.x; ux1 = t.y;
        } else {
            d2 t0 = *(const d2*)(x + 2 * (long)c), t1 = *(const d2*)(x + 2 * (long)c + 2);
            ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y;
        }
        const d2 ac = *(const d2*)(alpha + c);
        __syncthreads();
        for (int rr = wave; rr < 64; rr += 4) {
            const int i = row0 + rr;
            if (i >= n || c > i) continue;
            const double vx = rx[rr * DIM], vy = (DIM == 2) ? rx[rr * DIM + 1] : 0., ai = ra[rr];
            const d2 kv = *(const d2*)(Kinv + (long)i * ld + c);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = c + e;
                if (j > i || j >= n) continue;
                const double ux = e ? ux1 : ux0, uy = e ? uy1 : uy0;
                const double m = ai * (e ? ac.y : ac.x) - (e ? kv.y : kv.x);
                const double w = (j == i) ? 1. : 2.;
                const double E = rbf_exp(rbf_exp_arg<DIM>(ux, uy, vx, vy, kp, yinv), etab);
                const double d0 = ux - vx, d1 = uy - vy, mE = w * m * E;
                acc[0] += mE;
                acc[1] += mE * d0 * d0;
                acc[2] += mE * d0 * d1;
                acc[3] += mE * d1 * d1;
                if (j == i) acc[4] += m;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 5) part[(long)bl * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// out[k] = sum over workgroup partials, fixed order (one workgroup)
__global__ __launch_bounds__(256) void k_grad_final(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
    part += (long)blockIdx.x * nblocks * 8;    // one workgroup per object
    out += (long)blockIdx.x * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double red[4][5];
    double acc[5] = {0., 0., 0., 0., 0.};
    for (int b = tid; b < nblocks; b += 256) {
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += part[(long)b * 8 + k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 5) out[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------
// Measurement / self-test kernels
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters) {
    // bare v_mfma_f64_4x4x4_4b_f64 issue loop, 16 independent accumulators, varying operands
    double c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = 0.37 + 1e-3 * ((threadIdx.x * 7 + i * 13 + blockIdx.x) % 97);
        b[i] = -0.81 + 1e-3 * ((threadIdx.x * 11 + i * 5 + blockIdx.x) % 89);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = mfma4(a[i & 3], b[i >> 2], c[i]);
    }
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i];
    if (s == 12345.678) out[0] = s;
}

// streaming-store ceiling: every wave writes 4 KB contiguous per step (four 16-byte stores per lane in flight).  The
// one-store-per-lane grid-stride form reaches only 4.4 TB/s on a 1-GiB buffer (tools/asm_ablate.hip)
__global__ __launch_bounds__(256) void k_stream_store(double* out, long n2) {
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    d2 v = {1.0, 2.0};
    for (long b = w * 256; b + 255 < n2; b += nw * 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ((d2*)out)[b + q * 64 + lane] = v;
    }
}

// One 16 x 16 x 16 product in the register tiling every kernel uses: D[r][c] = C[r][c] + sum_k A[r][k] Bt[c][k]
// (A and Bt row-major 16 x 16 with k contiguous, like the LDS stages of K5), exact small integers.
// Block blk of the wave owns rows 4 blk + (0..3); column group J (cols 4 J + (0..3)) is the same in all
// blocks.  Also checks that a D block is a valid B operand of the same block (E_blk = A_blk D_blk).
__global__ void k_mfma_selftest(const double* A, const double* Bt, const double* Cin, double* D) {
    const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    double acc[4];
    for (int J = 0; J < 4; ++J) acc[J] = Cin[(4 * lb + lq) * 16 + 4 * J + lu];
    for (int kk = 0; kk < 4; ++kk)
        for (int J = 0; J < 4; ++J)
            acc[J] = mfma4(A[lr * 16 + lq + 4 * kk], Bt[(4 * J + lu) * 16 + lq + 4 * kk], acc[J]);
    for (int J = 0; J < 4; ++J) D[(4 * lb + lq) * 16 + 4 * J + lu] = acc[J];
    // chained product per block: E_blk (4 x 4) = Ablk (4 x 4, rows 4 blk.., cols 0..3 of A) * D_blk[:, cols of J = 0]
    double e = mfma4(A[(4 * lb + lu) * 16 + lq], acc[0], 0.);
    D[256 + lb * 16 + lq * 4 + lu] = e;
}
