// K3q: the 128 x 128 diagonal blocks of FOUR objects of a batch factored by ONE workgroup (round 4; gfx950, wave64, fp64).
//
// k_potrf_diag (kernels.hpp) owns a CU -- 150 KB of LDS -- for ~36 us per block with ONE wave on the critical chain (128 columns x
// ~300 cycles) and the other seven 85 % idle: for the 512-point light curves of BASELINE config[3] the four factorisations of an
// object are a quarter of its CU time.  A batch has many independent blocks, so here a workgroup takes four of them and runs
// their pivot chains SIDE BY SIDE, one per SIMD, as a two-level blocked factorisation in which only 64 x 64 quarters ever sit in
// LDS (4 x 33 KB):
//   phase 1   A11 = L11 L11^T            four 64 x 64 factorisations in lock-step: wave o is the pivot wave of object o (the chain of
//                                         potrf128_lds, 4 sub-steps of 16 columns), wave 4 + ((o + 2) & 3) its helper
//   phase 2   X21 = A21 L11^-T           the K4 solve, rows in registers, two 16-row strips per wave; L11 and its T^-T blocks in LDS
//             A22 -= X21 X21^T           X21 through LDS (over L11), ten 16 x 16 tiles per object, five per wave
//   phase 3   A22 = L22 L22^T            as phase 1
// Every entry receives exactly the operations it receives in potrf128_lds, in the same order (a trailing tile's sub-panel
// updates become one accumulator chain over ascending columns; the sub-panel below the first 64 columns is the panel solve of
// k_trsm_panel): same bits as k_potrf_diag (tests), same outputs (L in place, the eight T^-T blocks, log-determinant part, info).
// The T^-T blocks of a quarter live in the never-read UPPER tiles of its LDS image, so do the solve's wave-private tiles.
#pragma once
#include "kernels.hpp"

#define P4_LD 66                    // LDS row stride of a 64 x 64 quarter in doubles (528 B = 33 x 16 B: conflict-free like CGP_LDA)
#define P4_R (64 * P4_LD)           // doubles per object
#define P4_LC 136                   // column broadcast buffer of a pivot wave (2 x 64) + 2 reduction words, per object
#define P4_LDS_BYTES ((4 * P4_R + 4 * P4_LC) * 8)

// upper tiles of a quarter's image: T^-T block s -> (0,1) (0,2) (0,3) (1,2); wave-private tiles of the solve -> (1,3) (2,3)
__device__ __forceinline__ double* p4_wt(double* R, int s) {
    return s < 3 ? R + 16 * (s + 1) : R + 16 * P4_LD + 32;
}
__device__ __forceinline__ double* p4_xs(double* R, int which) { return R + (16 + 16 * which) * P4_LD + 48; }

// one 16 x 16 tile product, A and B tiles at row stride P4_LD (tile_mma_nt of kernels.hpp with the other stride)
template <bool NEG>
__device__ __forceinline__ void p4_tile_mma(double (&acc)[4], const double* Arow, const double* Brow, int lr, int u, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        d2 a = *(const d2*)(Arow + lr * P4_LD + 8 * t + 2 * q);
        if (NEG) a = -a;
        d2 b[4];
#pragma unroll
        for (int J = 0; J < 4; ++J) b[J] = *(const d2*)(Brow + (4 * J + u) * P4_LD + 8 * t + 2 * q);
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.x, b[J].x, acc[J]);
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.y, b[J].y, acc[J]);
    }
}

// phase (a) of a sub-step: the pivot wave factors the 16 x 16 tile at T (row stride P4_LD) in registers and leaves T^-T at Wt (row
// stride P4_LD).  The chain of potrf128_lds, operation for operation.  o16: index of the tile's first column inside the 128-block.
__device__ __forceinline__ int p4_pivot16(double* T, double* Wt, double* Lc, int o16, int bad) {
    const int lane = threadIdx.x & 63;
    double v[16], lprev = 0., pv;
    d2 bc[8];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double t = 0.;
        if (lane < 16) t = T[lane * P4_LD + c];
        else if (lane < 32) t = (c == lane - 16) ? 1. : 0.;
        v[c] = t;
    }
    pv = v[0];
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = readlane_f64(pv, j);
        if (!(d > 0.) && bad == 0) bad = o16 + j + 1;
        double l, sd;
        pivot_scale(d, v[j], l, sd);
        v[j] = (lane == j) ? sd : l;
        if (j > 0) {
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                const double lc = (c & 1) ? bc[c >> 1].y : bc[c >> 1].x;
                v[c] = __builtin_fma(-lprev, lc, v[c]);
            }
        }
        if (j < 15) {
            pv = __builtin_fma(-l, l, v[j + 1]);
            const double lc = readlane_f64(l, j + 1);
            v[j + 1] = __builtin_fma(-l, lc, v[j + 1]);
        }
        if (j < 14) {
            Lc[(j & 1) * 64 + lane] = l;
            wave_lds_fence();
#pragma unroll
            for (int c2 = (j + 2) >> 1; c2 < 8; ++c2) bc[c2] = *(const d2*)(Lc + (j & 1) * 64 + 2 * c2);
        }
        lprev = l;
    }
    __builtin_amdgcn_s_setprio(0);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[lane * P4_LD + c] = (c <= lane) ? v[c] : 0.;
    } else if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Wt[(lane - 16) * P4_LD + c] = (c >= lane - 16) ? v[c] : 0.;
    }
    return bad;
}

// The factorisation of the four 64 x 64 quarters in LDS, in lock-step (two workgroup barriers per sub-step).  `pivot`: this wave
// is the pivot wave of its object (R: the object's image, Lc: its broadcast buffer), else its helper; `on`: the object exists.
// o64: 0 / 64, the quarter's first column inside the 128-block.  Returns the updated `bad` (pivot waves).
__device__ __forceinline__ int p4_potrf64(double* R, double* Lc, bool pivot, bool on, int o64, int bad) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    for (int s = 0; s < 4; ++s) {
        const int o = s * 16;
        if (on && !pivot && s > 0) {
            // trailing tiles of sub-step s - 1: C[ti][tj] -= P_i P_j^T for 3 >= ti >= tj >= s, (ti, tj) != (s, s)
            const int po = (s - 1) * 16;
            for (int ti = s; ti < 4; ++ti)
                for (int tj = s; tj <= ti; ++tj) {
                    if (ti == s && tj == s) continue;
                    double* Ct = R + (ti * 16 + 4 * lb + lq) * P4_LD + tj * 16 + lu;
                    double acc[4];
#pragma unroll
                    for (int J = 0; J < 4; ++J) acc[J] = Ct[4 * J];
                    p4_tile_mma<true>(acc, R + (ti * 16) * P4_LD + po, R + (tj * 16) * P4_LD + po, lr, lu, lq);
#pragma unroll
                    for (int J = 0; J < 4; ++J) Ct[4 * J] = acc[J];
                }
        }
        if (on && pivot) bad = p4_pivot16(R + o * P4_LD + o, p4_wt(R, s), Lc, o64 + o, bad);
        __syncthreads();
        // sub-panel P[ti] = A[ti][s] T^-T: the pivot wave takes ti = s + 1 and then tile (s + 1, s + 1), the helper ti >= s + 2
        if (on) {
            const double* Wt = p4_wt(R, s);
            for (int ti = pivot ? s + 1 : s + 2; ti < (pivot ? min(s + 2, 4) : 4); ++ti) {
                double acc[4] = {0., 0., 0., 0.};
                const double* Arow = R + (ti * 16) * P4_LD + o;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    d2 a = *(const d2*)(Arow + lr * P4_LD + 8 * t + 2 * lq);
                    double b0[4], b1[4];
#pragma unroll
                    for (int J = 0; J < 4; ++J) {
                        b0[J] = Wt[(8 * t + 2 * lq) * P4_LD + 4 * J + lu];
                        b1[J] = Wt[(8 * t + 2 * lq + 1) * P4_LD + 4 * J + lu];
                    }
#pragma unroll
                    for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.x, b0[J], acc[J]);
#pragma unroll
                    for (int J = 0; J < 4; ++J) acc[J] = mfma4(a.y, b1[J], acc[J]);
                }
                wave_lds_fence();
#pragma unroll
                for (int J = 0; J < 4; ++J) R[(ti * 16 + 4 * lb + lq) * P4_LD + o + 4 * J + lu] = acc[J];
            }
            if (pivot && s < 3) {
                wave_lds_fence();
                const int t1 = s + 1;
                double* Ct = R + (t1 * 16 + 4 * lb + lq) * P4_LD + t1 * 16 + lu;
                double acc[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) acc[J] = Ct[4 * J];
                p4_tile_mma<true>(acc, R + (t1 * 16) * P4_LD + o, R + (t1 * 16) * P4_LD + o, lr, lu, lq);
#pragma unroll
                for (int J = 0; J < 4; ++J) Ct[4 * J] = acc[J];
            }
        }
        __syncthreads();
    }
    return bad;
}

// entry (r, c) and (r, c + 1) of the 128 x 128 block at (c0, c0) of object b: first touch (GEN) or loaded
template <int GEN>
__device__ __forceinline__ d2 p4_block2(const double* __restrict__ Mb, long ld, int c0, const GenSrc& g, int b, int r, int c, const double* etab) {
    if (GEN != 0) {
        constexpr int GD = GEN == 2 ? 2 : 1;
        const GenObj o = gen_obj<GD>(g, b);
        const GenRow rw = gen_row<GD>(g, o, c0 + r);
        const int cc = c0 + c;
        double ux0, uy0 = 0., ux1, uy1 = 0.;
        if (GD == 1) { const d2 t = *(const d2*)(o.x + cc); ux0 = t.x; ux1 = t.y; }
        else { const d2 t0 = *(const d2*)(o.x + 2 * (long)cc), t1 = *(const d2*)(o.x + 2 * (long)cc + 2); ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y; }
        d2 v;
        v.x = gen_val<GD>(o, rw, c0 + r, cc, ux0, uy0, etab);
        v.y = gen_val<GD>(o, rw, c0 + r, cc + 1, ux1, uy1, etab);
        return v;
    }
    return *(const d2*)(Mb + (long)(c0 + r) * ld + c0 + c);
}

template <int GEN>
__global__ __launch_bounds__(512) void k_potrf_diag4(
    double* __restrict__ M, long ld, long sM, int c0, double* __restrict__ wsW, long sW,
    double* __restrict__ logdet_parts, long sLd, int* __restrict__ info, int blk, GenSrc g, int B) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ __attribute__((aligned(16))) double etab[GEN != 0 ? 128 : 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    if (GEN != 0) {
        rbf_stage_table(etab);
        __syncthreads();
    }
    // ---- phase 0: A11 of the four objects into LDS (tiles on and below the diagonal)
    for (int o = 0; o < 4; ++o) {
        const int b = 4 * blockIdx.x + o;
        if (b >= B) continue;
        const double* Mb = M + (long)b * sM;
        double* R = smem + o * P4_R;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 512, r = idx >> 5, c = (idx & 31) * 2;
            if ((c >> 4) <= (r >> 4)) *(d2*)(R + r * P4_LD + c) = p4_block2<GEN>(Mb, ld, c0, g, b, r, c, etab);
        }
    }
    __syncthreads();
    // roles of the factorisations: wave o < 4 is the pivot wave of object o (SIMD o), wave w >= 4 the helper of object (w - 2) & 3
    const bool pivot = wave < 4;
    const int fo = pivot ? wave : ((wave - 2) & 3);
    const bool fon = 4 * (int)blockIdx.x + fo < B;
    double* FR = smem + fo * P4_R;
    double* FLc = smem + 4 * P4_R + fo * P4_LC;
    int bad = p4_potrf64(FR, FLc, pivot, fon, 0, 0);
    // ---- phase 2: this wave's object for the solve and the update: o = wave & 3, half = wave >> 2
    const int so = wave & 3, half = wave >> 2;
    const int sb = 4 * blockIdx.x + so;
    const bool son = sb < B;
    double* SR = smem + so * P4_R;
    double* Mb = M + (long)sb * sM;
    double* Tg = Mb + (long)c0 * ld + c0;                   // the block in HBM
    double* wsb = wsW + (long)sb * sW + (long)blk * (8 * 256);
    double lg0 = 0.;
    if (son) {
        // write-back of L11 (lower tiles) and of the T^-T blocks 0 .. 3: the object's two waves share the rows
        for (int k = 0; k < 16; ++k) {
            const int idx = lane + 64 * (half + 2 * k), r = idx >> 5, c = (idx & 31) * 2;       // 2048 pairs: 64 rows x 32
            if ((c >> 4) <= (r >> 4)) *(d2*)(Tg + (long)r * ld + c) = *(const d2*)(SR + r * P4_LD + c);
        }
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * (half + 2 * k), s = idx >> 7, e = idx & 127, r = e >> 3, c = (e & 7) * 2;   // 4 blocks x 128 pairs
            *(d2*)(wsb + s * 256 + r * 16 + c) = *(const d2*)(p4_wt(SR, s) + r * P4_LD + c);
        }
        if (half == 0) {
            // 2 log L_ii of rows 0 .. 63 in one wave: the tree of k_potrf_diag's wave 0
            lg0 = 2. * log(SR[lane * P4_LD + lane]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lg0 += __shfl_down(lg0, off);
        }
    }
    double XR[2][4][4];
    if (son) {
        double* Xs = p4_xs(SR, half);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int row = 64 + 16 * (2 * half + st) + 4 * lb + lq;
            double (&S)[4][4] = XR[st];                      // solved in place: S[jb] becomes X_jb
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const d2 t0 = p4_block2<GEN>(Mb, ld, c0, g, sb, row, jb * 16 + 4 * lu, etab), t1 = p4_block2<GEN>(Mb, ld, c0, g, sb, row, jb * 16 + 4 * lu + 2, etab);
                S[jb][0] = t0.x; S[jb][1] = t0.y; S[jb][2] = t1.x; S[jb][3] = t1.y;
            }
            double* Ap = Tg + (long)row * ld + 4 * lu;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                wave_lds_fence();
                *(d2*)(Xs + (4 * lb + lq) * P4_LD + 4 * lu) = (d2){S[jb][0], S[jb][1]};
                *(d2*)(Xs + (4 * lb + lq) * P4_LD + 4 * lu + 2) = (d2){S[jb][2], S[jb][3]};
                wave_lds_fence();
                double x[4] = {0., 0., 0., 0.};
                const double* Wt = p4_wt(SR, jb);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    d2 a = *(const d2*)(Xs + lr * P4_LD + 8 * t + 2 * lq);
                    const double* w0 = Wt + (8 * t + 2 * lq) * P4_LD + 4 * lu;
                    d2 p0 = *(const d2*)(w0), p1 = *(const d2*)(w0 + 2), p2 = *(const d2*)(w0 + P4_LD), p3 = *(const d2*)(w0 + P4_LD + 2);
                    x[0] = mfma4(a.x, p0.x, x[0]); x[1] = mfma4(a.x, p0.y, x[1]);
                    x[2] = mfma4(a.x, p1.x, x[2]); x[3] = mfma4(a.x, p1.y, x[3]);
                    x[0] = mfma4(a.y, p2.x, x[0]); x[1] = mfma4(a.y, p2.y, x[1]);
                    x[2] = mfma4(a.y, p3.x, x[2]); x[3] = mfma4(a.y, p3.y, x[3]);
                }
                *(d2*)(Ap + jb * 16) = (d2){x[0], x[1]};
                *(d2*)(Ap + jb * 16 + 2) = (d2){x[2], x[3]};
#pragma unroll
                for (int J = 0; J < 4; ++J) S[jb][J] = x[J];
                if (jb < 3) {
                    wave_lds_fence();
                    *(d2*)(Xs + (4 * lb + lq) * P4_LD + 4 * lu) = (d2){x[0], x[1]};
                    *(d2*)(Xs + (4 * lb + lq) * P4_LD + 4 * lu + 2) = (d2){x[2], x[3]};
                    wave_lds_fence();
                    d2 a[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) a[t] = -*(const d2*)(Xs + lr * P4_LD + 8 * t + 2 * lq);
#pragma unroll
                    for (int i = jb + 1; i < 4; ++i) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            d2 bb[4];
#pragma unroll
                            for (int J = 0; J < 4; ++J)
                                bb[J] = *(const d2*)(SR + (i * 16 + 4 * lu + J) * P4_LD + jb * 16 + 8 * t + 2 * lq);
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
    __syncthreads();                                         // every wave is done with L11 and its T^-T blocks
    if (son) {
        // X21 into LDS (over L11): rows 16 (2 half + st) .., all 64 columns
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                double* dst = SR + (16 * (2 * half + st) + 4 * lb + lq) * P4_LD + jb * 16 + 4 * lu;
                *(d2*)(dst) = (d2){XR[st][jb][0], XR[st][jb][1]};
                *(d2*)(dst + 2) = (d2){XR[st][jb][2], XR[st][jb][3]};
            }
    }
    __syncthreads();
    // A22 -= X21 X21^T: the ten lower 16 x 16 tiles of the object, five per wave (tile t = 5 half + n: row-major over the triangle)
    double acc5[5][4];
    if (son) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int t = 5 * half + n;
            const int ti = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3)), tj = t - ti * (ti + 1) / 2;
            // (potrf128_lds' accumulator layout: acc[J] <-> row 4 lb + lq, column lu + 4 J of the tile; loaded pairwise)
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const int r = 64 + 16 * ti + 4 * lb + lq, c = 64 + 16 * tj + lu + 4 * J;
                const d2 v = p4_block2<GEN>(Mb, ld, c0, g, sb, r, c & ~1, etab);
                acc5[n][J] = (c & 1) ? v.y : v.x;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                p4_tile_mma<true>(acc5[n], SR + (ti * 16) * P4_LD + 16 * kk, SR + (tj * 16) * P4_LD + 16 * kk, lr, lu, lq);
        }
    }
    __syncthreads();                                         // every wave has read X21
    if (son) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int t = 5 * half + n;
            const int ti = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3)), tj = t - ti * (ti + 1) / 2;
            double* Ct = SR + (ti * 16 + 4 * lb + lq) * P4_LD + tj * 16 + lu;
#pragma unroll
            for (int J = 0; J < 4; ++J) Ct[4 * J] = acc5[n][J];
        }
    }
    __syncthreads();
    // ---- phase 3: A22 = L22 L22^T
    const int bad2 = p4_potrf64(FR, FLc, pivot, fon, 64, 0);
    if (bad == 0) bad = bad2;
    if (pivot && fon && lane == 0) ((int*)(FLc + 129))[0] = bad;      // the pivot wave's verdict for the write-back
    __syncthreads();
    if (son) {
        double* T22 = Tg + (long)64 * ld + 64;
        for (int k = 0; k < 16; ++k) {
            const int idx = lane + 64 * (half + 2 * k), r = idx >> 5, c = (idx & 31) * 2;
            if ((c >> 4) <= (r >> 4)) *(d2*)(T22 + (long)r * ld + c) = *(const d2*)(SR + r * P4_LD + c);
        }
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * (half + 2 * k), s = idx >> 7, e = idx & 127, r = e >> 3, c = (e & 7) * 2;
            *(d2*)(wsb + (4 + s) * 256 + r * 16 + c) = *(const d2*)(p4_wt(SR, s) + r * P4_LD + c);
        }
        if (half == 0) {
            double lg1 = 2. * log(SR[lane * P4_LD + lane]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lg1 += __shfl_down(lg1, off);
            if (lane == 0) {
                logdet_parts[(long)sb * sLd + blk] = lg0 + lg1;
                const int bd = ((const int*)(smem + 4 * P4_R + so * P4_LC + 129))[0];
                if (bd != 0 && info[sb] == 0) info[sb] = c0 + bd;
            }
        }
    }
}
