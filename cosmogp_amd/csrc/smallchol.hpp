// Left-looking panel kernels for batches of SMALL objects (round 4; gfx950, wave64, fp64).
//
// The launch-per-step route of kernels.hpp factors an object of a few 128-column panels recursively: panel solve, update of the
// right half, recurse.  For the 512-point light curves of BASELINE config[3] that is 3 update launches of 64 x 64 tiles whose K
// loops (8 or 16 stages of 32 MFMAs per wave) are bound by the latency of their stages, and every updated tile makes a round
// trip through HBM before its panel solve reads it again.  Here panel p >= 1 of an object is produced LEFT-LOOKING by two
// kinds of workgroup, each with a 128-wide tile in registers from its first touch to its final value:
//   k_ll_diag   tile (p, p) = [generated | loaded] - sum_{k < p} L(p, k) L(p, k)^T   (one K loop of depth 128 p, the update tile of
//               dag.hpp: 8 waves of 64 x 32, 128 MFMAs per wave and stage), then the factorisation in LDS (potrf128_lds);
//   k_ll_slab   a slab of 128 (or 64: the tall rows) rows below: S = [generated | loaded] - sum_{k < p} L(i, k) L(p, k)^T in
//               registers, through LDS into the panel solve's layout, X = S L_pp^-T (the K4 code), written ONCE.
// Panel 0 has nothing to its left: k_potrf_diag[_gen] and k_trsm_panel do it as before.  Per fit of a 512-point object
// with 19 prediction points: 11 launches instead of 12, no update tile ever re-read, 5.6 -> ~3.6 MB of HBM traffic.
// The sums of an entry run over the columns to its left in ascending order in one accumulator chain (the recursive route
// splits them by halves): results differ from that route in the last bits (<= 1e-13 relative on every output, tests), and
// are bit-identical between first-touch generation and assembled input, single fits and batches, any chunking.
#pragma once
#include "kernels.hpp"

#define LL_THREADS 512
#define LL_KS 20                   // LDS row stride of a stage in doubles (conflict-free fragment reads: dag.hpp)
#define LL_LDS_DOUBLES (CGP_NB * CGP_LDA + 8 * 256 + 160)
#define LL_LDS_BYTES (LL_LDS_DOUBLES * 8)

// ---------------------------------------------------------------------------------------------------------------
// acc (+)= A (32 MT rows x K at Ag) B^T (128 rows x K at Bg), both row-major with leading dimension ld, K a multiple of 32.
// 8 waves as 2 x 4, wave tile (16 MT) x 32; KB = 16 stages double-buffered in LDS through registers, global loads two stages
// ahead, the MFMA fragments of half a stage read while the MFMAs of the half before issue (the loop of dag_upd_tile).
//   acc[i][2 jj], acc[i][2 jj + 1] <-> row (16 MT) wm + 16 i + 4 lb + lq, columns 32 wn + 8 jj + 2 lp + (0, 1)
// wave_on = false: this wave's tile is not wanted (it still stages and keeps the barriers).  arows: rows of A that carry data
// (the others are not loaded: zeros in the stage).
// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__device__ __forceinline__ void ll_kloop(double (&acc)[MT][8], const double* __restrict__ Ag, const double* __restrict__ Bg, long ld,
                                         int K, double* smem, bool wave_on, int arows) {
    constexpr int TM = 32 * MT, TN = 128, WN = 4, KB = 16, KS = LL_KS, CPR = KB / 2, NJ = 8;
    constexpr int PA = TM * CPR / LL_THREADS, PB = TN * CPR / LL_THREADS, RSTEP = LL_THREADS / CPR;      // 64 rows per staging pass
    static_assert(PA >= 1, "tile too small");
    double* As = smem;                      // [2][TM * KS]
    double* Bs = smem + 2 * TM * KS;        // [2][TN * KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    const int srow = tid / CPR, sch = (tid % CPR) * 2;
    d2 ra[2][PA], rb[2][PB];
    const int nk = K / KB;
    auto gload = [&](auto par, int kt) {
        constexpr int P = decltype(par)::value;
        const long k1 = (long)kt * KB + sch;
#pragma unroll
        for (int q = 0; q < PA; ++q)
            ra[P][q] = (srow + RSTEP * q < arows) ? *(const d2*)(Ag + (long)(srow + RSTEP * q) * ld + k1) : (d2){0., 0.};
#pragma unroll
        for (int q = 0; q < PB; ++q) rb[P][q] = *(const d2*)(Bg + (long)(srow + RSTEP * q) * ld + k1);
    };
    auto lstore = [&](auto par, int buf) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int q = 0; q < PA; ++q) *(d2*)(&As[buf * TM * KS + (srow + RSTEP * q) * KS + sch]) = ra[P][q];
#pragma unroll
        for (int q = 0; q < PB; ++q) *(d2*)(&Bs[buf * TN * KS + (srow + RSTEP * q) * KS + sch]) = rb[P][q];
    };
    d2 fa[2][MT], fb[2][NJ];
    auto fread = [&](auto setc, int buf, int t) {
        constexpr int S = decltype(setc)::value;
        if (!wave_on) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            fa[S][i] = *(const d2*)(&As[buf * TM * KS + (wm * (16 * MT) + i * 16 + lr) * KS + 8 * t + 2 * lq]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[S][j] = *(const d2*)(&Bs[buf * TN * KS + (wn * 32 + 8 * (j >> 1) + 2 * lp + (j & 1)) * KS + 8 * t + 2 * lq]);
    };
    auto fmma = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        if (!wave_on) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[S][i].x, fb[S][j].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[S][i].y, fb[S][j].y, acc[i][j]);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (nk == 0) return;
    gload(P0{}, 0);
    gload(P1{}, 1);
    __builtin_amdgcn_sched_barrier(0);
    lstore(P0{}, 0);
    __syncthreads();
    fread(P0{}, 0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // stage in buffer 0; stage kt + 1 goes to buffer 1 meanwhile
        gload(P0{}, min(kt + 2, nk - 1));
        fread(P1{}, 0, 1);
        fmma(P0{});
        lstore(P1{}, 1);
        __syncthreads();                 // buffer 1 complete; every wave's reads of buffer 0 have returned
        fread(P0{}, 1, 0);
        fmma(P1{});
        // stage in buffer 1; stage kt + 2 goes to buffer 0 meanwhile
        gload(P1{}, min(kt + 3, nk - 1));
        fread(P1{}, 1, 1);
        fmma(P0{});
        lstore(P0{}, 0);
        __syncthreads();
        fread(P0{}, 0, 0);
        fmma(P1{});
    }
    __syncthreads();                     // (every wave is done with the stage buffers: the caller reuses the LDS)
}

// The tile's start values, negated (the accumulators run on -C like dag_upd_tile: C - sum A B^T = -(-C + sum A B^T)):
// first touch (GEN: computed, kernels.hpp GenSrc) or loaded.  row0 / col0: absolute row / column of the tile's origin.
template <int MT, int GEN>
__device__ __forceinline__ void ll_init_acc(double (&acc)[MT][8], const double* __restrict__ Cg, long ld, const GenSrc& g, int b,
                                            int row0, int col0, const double* etab, bool wave_on) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / 4, wn = wave % 4;
    const int lq = lane >> 4, lp = lane & 3, lb = (lane >> 2) & 3;
    const int r_ = wm * (16 * MT) + 4 * lb + lq, c_ = wn * 32 + 2 * lp;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.;
    if (!wave_on) return;
    if (GEN != 0) {
        constexpr int GD = GEN == 2 ? 2 : 1;
        const GenObj o = gen_obj<GD>(g, b);
        double ux[8], uy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = col0 + c_ + 8 * (j >> 1) + (j & 1);
            ux[j] = o.x[(long)c * GD];
            uy[j] = GD == 2 ? o.x[(long)c * GD + 1] : 0.;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = row0 + r_ + 16 * i;
            const GenRow rw = gen_row<GD>(g, o, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = -gen_val<GD>(o, rw, r, col0 + c_ + 8 * (j >> 1) + (j & 1), ux[j], uy[j], etab);
        }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const d2 v = *(const d2*)(Cg + (long)(r_ + 16 * i) * ld + c_ + 8 * jj);
                acc[i][2 * jj] = -v.x;
                acc[i][2 * jj + 1] = -v.y;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ll_diag: diagonal tile of panel p = c0 / 128 >= 1 of every object of the launch (grid (1, 1, B)), see the head of the file.
// Wave tiles entirely above the diagonal (wm = 0, wn >= 2) sit the K loop out.
// ---------------------------------------------------------------------------------------------------------------
template <int GEN>
__global__ __launch_bounds__(LL_THREADS) void k_ll_diag(
    double* __restrict__ M, long ld, long sM, int c0, double* __restrict__ wsW, long sW,
    double* __restrict__ logdet_parts, long sLd, int* __restrict__ info, int blk, GenSrc g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ __attribute__((aligned(16))) double etab[GEN != 0 ? 128 : 2];
    double* As = smem;                          // [128][CGP_LDA] (after the K loop, whose stage buffers share the space)
    double* Ws = smem + CGP_NB * CGP_LDA;
    double* Lc = Ws + 8 * 256;
    const int b = blockIdx.z;
    M += (long)b * sM;
    wsW += (long)b * sW + (long)blk * (8 * 256);
    double* T = M + (long)c0 * ld + c0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 4, wn = wave % 4, lq = lane >> 4, lp = lane & 3, lb = (lane >> 2) & 3;
    if (GEN != 0) {
        rbf_stage_table(etab);
        __syncthreads();
    }
    const bool wave_on = !(wm == 0 && wn >= 2);
    double acc[4][8];
    ll_init_acc<4, GEN>(acc, T, ld, g, b, c0, c0, etab, wave_on);
    ll_kloop<4>(acc, M + (long)c0 * ld, M + (long)c0 * ld, ld, c0, smem, wave_on, 128);
    if (wave_on) {
        const int r_ = wm * 64 + 4 * lb + lq, c_ = wn * 32 + 2 * lp;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *(d2*)(As + (r_ + 16 * i) * CGP_LDA + c_ + 8 * jj) = (d2){-acc[i][2 * jj], -acc[i][2 * jj + 1]};
    }
    __syncthreads();
    const int bad = potrf128_lds(As, Ws, Lc);
    // write L back (tiles on and below the block diagonal), T^-T blocks, logdet, info (the tail of k_potrf_diag)
    {
        constexpr int NCH = CGP_NB * (CGP_NB / 2) / LL_THREADS;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = tid + k * LL_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            if ((c >> 4) <= (r >> 4)) *(d2*)(T + (long)r * ld + c) = *(const d2*)(As + r * CGP_LDA + c);
        }
    }
    for (int idx = tid; idx < 8 * 256; idx += LL_THREADS) wsW[idx] = Ws[idx];
    double lg = 0.;
    if (tid < CGP_NB) lg = 2. * log(As[tid * CGP_LDA + tid]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    double* red = Lc + 128;
    if (lane == 0 && wave < 2) red[wave] = lg;
    __syncthreads();
    if (tid == 0) {
        logdet_parts[(long)b * sLd + blk] = red[0] + red[1];
        if (bad != 0 && info[b] == 0) info[b] = c0 + bad;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One slab of one panel: rows [row0, row0 + TM) (the first `valid` carry data), panel p = c0 / 128, TM = 32 MT.
//   S = [generated | loaded] - sum_{k < p} L(rows, k) L(p, k)^T in registers (one K loop of depth c0), through LDS into the
//   panel solve's layout (wave w: rows 16 w .. 16 w + 15, all 128 columns), L_pp + its T^-T blocks staged, X = S L_pp^-T
//   (the K4 code: kernels.hpp k_trsm_panel), written once.  Every thread of the workgroup calls this; no thread leaves early
//   (so that a caller may walk several panels in a loop).
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int GEN>
__device__ __forceinline__ void ll_slab_body(double* __restrict__ M, long ld, int c0, int row0, int valid,
                                             const double* __restrict__ wsW, const GenSrc& g, int b, double* smem, const double* etab) {
    constexpr int TM = 32 * MT;
    double* Ls = smem;                               // S tile [TM][CGP_LDA], then L_pp [128][CGP_LDA] (strictly-lower tiles)
    double* Ws = smem + CGP_NB * CGP_LDA;            // [8][16][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 4, wn = wave % 4;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3, lp = lu;
    const bool wave_on = wm * (16 * MT) < valid;
    double acc[MT][8];
    ll_init_acc<MT, GEN>(acc, M + (long)row0 * ld + c0, ld, g, b, row0, c0, etab, wave_on);
    ll_kloop<MT>(acc, M + (long)row0 * ld, M + (long)c0 * ld, ld, c0, smem, wave_on, valid);
    // S = -acc through LDS into the panel solve's layout: wave w owns rows 16 w .. 16 w + 15, all 128 columns
    if (wave_on) {
        const int r_ = wm * (16 * MT) + 4 * lb + lq, c_ = wn * 32 + 2 * lp;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *(d2*)(Ls + (r_ + 16 * i) * CGP_LDA + c_ + 8 * jj) = (d2){-acc[i][2 * jj], -acc[i][2 * jj + 1]};
    }
    __syncthreads();
    const bool has_rows = wave * 16 < valid && wave * 16 < TM;
    double S[8][4];
    if (has_rows) {
        const double* sp = Ls + (wave * 16 + 4 * lb + lq) * CGP_LDA + 4 * lu;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            const d2 t0 = *(const d2*)(sp + jb * 16), t1 = *(const d2*)(sp + jb * 16 + 2);
            S[jb][0] = t0.x; S[jb][1] = t0.y; S[jb][2] = t1.x; S[jb][3] = t1.y;
        }
    }
    __syncthreads();
    // L_pp (strictly-lower 16 x 16 tiles: rows 16 .. 127) and its T^-T blocks -> LDS
    {
        const double* Lg = M + (long)c0 * ld + c0;
        constexpr int NCH = 112 * 64 / LL_THREADS;          // 14
        d2 v[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int idx = tid + q * LL_THREADS, r = 16 + (idx >> 6), c = (idx & 63) * 2;
            v[q] = (c >> 4) < (r >> 4) ? *(const d2*)(Lg + (long)r * ld + c) : (d2){0., 0.};
        }
        d2 w[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) w[q] = *(const d2*)(wsW + 2 * (tid + q * LL_THREADS));
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int idx = tid + q * LL_THREADS, r = 16 + (idx >> 6), c = (idx & 63) * 2;
            if ((c >> 4) < (r >> 4)) *(d2*)(Ls + r * CGP_LDA + c) = v[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) *(d2*)(Ws + 2 * (tid + q * LL_THREADS)) = w[q];
    }
    __syncthreads();
    if (has_rows) {
    // X = S L_pp^-T, right-looking over the eight 16-column tiles (the K4 code: kernels.hpp k_trsm_panel)
    const int xg = wave < 6 ? 0 : 1, xt = wave - 6 * xg;
    double* Xs = Ls + xg * 16 * CGP_LDA + 16 + 18 * xt;
    double* Ap = M + (long)(row0 + wave * 16 + 4 * lb + lq) * ld + c0 + 4 * lu;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        wave_lds_fence();
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu) = (d2){S[jb][0], S[jb][1]};
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu + 2) = (d2){S[jb][2], S[jb][3]};
        wave_lds_fence();
        double x[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d2 a = *(const d2*)(Xs + lr * CGP_LDA + 8 * t + 2 * lq);
            const double* w0 = Ws + jb * 256 + (8 * t + 2 * lq) * 16 + 4 * lu;
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
    }
}

// k_ll_slab: rows [r0 + TM bx, + TM) of panel p = c0 / 128 >= 1, TM = 32 MT (128: matrix rows, 64: tall rows), grid
// (slabs, 1, B) or its XCD-affine form.  nrows: rows of the launch that carry data (a last slab may be part empty).
template <int MT, int GEN>
__global__ __launch_bounds__(LL_THREADS) void k_ll_slab(
    double* __restrict__ M, long ld, long sM, int c0, int r0, const double* __restrict__ wsW, long sW, int blk, int nrows,
    GenSrc g, int xcdB) {
    constexpr int TM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ __attribute__((aligned(16))) double etab[GEN != 0 ? 128 : 2];
    int b = blockIdx.z, bx = blockIdx.x;
    if (xcdB > 0 && !xcd_affine(b, bx, xcdB)) return;
    M += (long)b * sM;
    wsW += (long)b * sW + (long)blk * (8 * 256);
    if (GEN != 0) {
        rbf_stage_table(etab);
        __syncthreads();
    }
    ll_slab_body<MT, GEN>(M, ld, c0, r0 + bx * TM, min(TM, nrows - bx * TM), wsW, g, b, smem, etab);
}
