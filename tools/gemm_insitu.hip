// Developer harness: replay the trailing updates of one N x N (+ tall rows) recursive factorisation with the
// fit's own pointer layout and time them per update depth K, for several tile forms.
//   usage: gemm_insitu [N=8192] [tall=2560]
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#include <cstring>
#include <unistd.h>
// core-clock monitor: one lane samples (core cycles, 100 MHz wall ticks) every ~20 us until *stop != 0
__global__ void k_clockmon(volatile int* stop, long long* out, int maxn) {
    if (threadIdx.x != 0) return;
    int n = 0;
    long long c0 = clock64(), w0 = wall_clock64();
    while (n < maxn && *stop == 0) {
        long long w;
        do { w = wall_clock64(); } while (w - w0 < 2000);
        long long c = clock64();
        out[2 * n] = c - c0; out[2 * n + 1] = w - w0;
        c0 = c; w0 = w; ++n;
        __builtin_amdgcn_s_sleep(8);
    }
    out[2 * maxn] = n;
}
// (developer experiment, kept out of the product header: measured no faster than the register-staged kernel)
// ------------------------------------------------------------------------------------
// K5-dl: the same update with DIRECT global -> LDS loads (global_load_lds_dwordx4, gfx950): no staging
// registers, no ds_write; a stage of the A / B panels lands in LDS as the lanes deliver it, 1 KiB per wave
// instruction = 8 rows x 128 B.  Rows are unpadded (16 doubles); the 16-byte chunk c of row r is kept at
// position c ^ (r & 7), which the loading lane realises by fetching the global chunk (lane & 7) ^ (lane >> 3) --
// the fragment reads of 8 consecutive rows then hit 8 different bank groups.  Same tile map, same MFMA order
// (bit-identical results) as k_gemm_nt<TM, TN, WM, WN, 16>.
// ------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_nt_dl(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int tile_skip, int ktri,
    int tile_end = 0) {
    constexpr int KB = 16, NW = WM * WN;
    constexpr int MT = TM / WM / 16, NJ = TN / WN / 4;
    constexpr int GA = TM / 8 / NW, GB = TN / 8 / NW;        // 8-row groups per wave per stage
    static_assert(GA >= 1 && GB >= 1 && TM % (8 * NW) == 0 && TN % (8 * NW) == 0, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(128))) double As[2][TM * KB];
    __shared__ __attribute__((aligned(128))) double Bs[2][TN * KB];
    const int bz = blockIdx.z;
    C += (long)bz * sC;
    A += (long)bz * sAB;
    B += (long)bz * sAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    constexpr int RT = TM / TN;
    static_assert(TM == RT * TN && (RT == 1 || RT == 2), "tile shape");
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot_all = ntri + (nr - (tri ? nt : 0)) * nc;
    const int ntot = tile_end > 0 ? min(tile_end, ntot_all) : ntot_all;
    // loading role of this lane: row (lane >> 3) of an 8-row group, LDS slot (lane & 7), global chunk below
    const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
    for (int bid = blockIdx.x + tile_skip; bid < ntot; bid += gridDim.x) {
    int ti, tj;
    if (bid < ntri) {
        int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
        while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
        while (RT * i * (i + 1) / 2 > bid) --i;
        ti = i;
        tj = bid - RT * i * (i + 1) / 2;
    } else {
        const int r = bid - ntri;
        ti = (tri ? nt : 0) + r / nc;
        tj = r % nc;
    }
    const int kstart = ktri ? ti * TM : 0;
    const double* Ag = A + (long)(ti * TM) * ld + kstart;
    const double* Bg = B + (long)(tj * TN) * ld + kstart;
    const int nk = (K - kstart) / KB;
    const double* pa[GA];
    const double* pb[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) pa[g] = Ag + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
#pragma unroll
    for (int g = 0; g < GB; ++g) pb[g] = Bg + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;

    double acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.;

    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int g = 0; g < GA; ++g)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[g] + (long)kt * KB),
                                             (__attribute__((address_space(3))) void*)(&As[buf][8 * (g * NW + wave) * KB]), 16, 0, 0);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[g] + (long)kt * KB),
                                             (__attribute__((address_space(3))) void*)(&Bs[buf][8 * (g * NW + wave) * KB]), 16, 0, 0);
    };
    auto compute = [&](int cur) {
        d2 fa[MT], fb[NJ];
#pragma unroll
        for (int t = 0; t < KB / 8; ++t) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wm * (TM / WM) + i * 16 + lr;
                fa[i] = *(const d2*)(&As[cur][row * KB + 2 * ((4 * t + lq) ^ (row & 7))]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = wn * (TN / WN) + j * 4 + lp;
                fb[j] = *(const d2*)(&Bs[cur][row * KB + 2 * ((4 * t + lq) ^ (row & 7))]);
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
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt+1 goes into the buffer every wave finished reading before the barrier that ended stage kt-1
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    double* Cg = C + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN + wn * (TN / WN) + lp;
    if (beta != 0.) {
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


// Software-pipelined variant (experiment): 3 LDS stages filled by direct loads two stages ahead, fragments of the
// next half-stage read from LDS while the MFMAs of the current half issue, ONE barrier per stage placed between
// the two halves.  128 x 64 tile, 8 waves (GA + GB = 3 loads per lane and stage).
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_nt_sp(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int tile_skip, int ktri,
    int tile_end = 0) {
    constexpr int KB = 16, NW = WM * WN, NBUF = 3;
    constexpr int MT = TM / WM / 16, NJ = TN / WN / 4;
    constexpr int GA = TM / 8 / NW, GB = TN / 8 / NW;
    static_assert(GA + GB == 3, "wait counts below assume 3 loads per lane and stage");
    __shared__ __attribute__((aligned(128))) double As[NBUF][TM * KB];
    __shared__ __attribute__((aligned(128))) double Bs[NBUF][TN * KB];
    const int bz = blockIdx.z;
    C += (long)bz * sC; A += (long)bz * sAB; B += (long)bz * sAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    constexpr int RT = TM / TN;
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot_all = ntri + (nr - (tri ? nt : 0)) * nc;
    const int ntot = tile_end > 0 ? min(tile_end, ntot_all) : ntot_all;
    const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
    for (int bid = blockIdx.x + tile_skip; bid < ntot; bid += gridDim.x) {
    int ti, tj;
    if (bid < ntri) {
        int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
        while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
        while (RT * i * (i + 1) / 2 > bid) --i;
        ti = i; tj = bid - RT * i * (i + 1) / 2;
    } else {
        const int r = bid - ntri;
        ti = (tri ? nt : 0) + r / nc; tj = r % nc;
    }
    const int kstart = ktri ? ti * TM : 0;
    const double* Ag = A + (long)(ti * TM) * ld + kstart;
    const double* Bg = B + (long)(tj * TN) * ld + kstart;
    const int nk = (K - kstart) / KB;
    const double* pa[GA];
    const double* pb[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) pa[g] = Ag + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
#pragma unroll
    for (int g = 0; g < GB; ++g) pb[g] = Bg + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
    double acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.;
    // fragment offsets (doubles) inside a stage, per half t
    int oa[2][MT], ob[2][NJ];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < MT; ++i) { const int row = wm * (TM / WM) + i * 16 + lr; oa[t][i] = row * KB + 2 * ((4 * t + lq) ^ (row & 7)); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int row = wn * (TN / WN) + j * 4 + lp; ob[t][j] = row * KB + 2 * ((4 * t + lq) ^ (row & 7)); }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&As[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pa[g] + (long)kt * KB) : "memory");
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&Bs[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pb[g] + (long)kt * KB) : "memory");
        }
    };
    auto frags = [&](int buf, int t, d2* fa, d2* fb) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *(const d2*)(&As[buf][oa[t][i]]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = *(const d2*)(&Bs[buf][ob[t][j]]);
    };
    auto mma = [&](const d2* fa, const d2* fb) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[i].x, fb[j].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[i].y, fb[j].y, acc[i][j]);
    };
    d2 fa0[MT], fb0[NJ], fa1[MT], fb1[NJ];
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    frags(0, 0, fa0, fb0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int b1 = buf == 2 ? 0 : buf + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        if (kt + 2 < nk) issue(kt + 2, b2);
        frags(buf, 1, fa1, fb1);
        mma(fa0, fb0);
        if (kt + 1 < nk) {
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            frags(b1, 0, fa0, fb0);
        }
        mma(fa1, fb1);
        buf = b1;
    }
    __syncthreads();        // the next tile's prologue overwrites stage buffers
    double* Cg = C + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN + wn * (TN / WN) + lp;
    if (beta != 0.) {
        double cv[MT][NJ];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) cv[i][j] = Cg[(long)(i * 16) * ldc + j * 4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j] + beta * cv[i][j];
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j];
    }
    }
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_nt_sp2(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int tile_skip, int ktri,
    int tile_end = 0) {
    constexpr int KB = 16, NW = WM * WN, NBUF = 3;
    constexpr int MT = TM / WM / 16, NJ = TN / WN / 4;
    constexpr int GA = TM / 8 / NW, GB = TN / 8 / NW;
    static_assert(GA + GB == 3, "wait counts below assume 3 loads per lane and stage");
    __shared__ __attribute__((aligned(128))) double As[NBUF][TM * KB];
    __shared__ __attribute__((aligned(128))) double Bs[NBUF][TN * KB];
    const int bz = blockIdx.z;
    C += (long)bz * sC; A += (long)bz * sAB; B += (long)bz * sAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    constexpr int RT = TM / TN;
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot_all = ntri + (nr - (tri ? nt : 0)) * nc;
    const int ntot = tile_end > 0 ? min(tile_end, ntot_all) : ntot_all;
    const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
    for (int bid = blockIdx.x + tile_skip; bid < ntot; bid += gridDim.x) {
    int ti, tj;
    if (bid < ntri) {
        int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
        while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
        while (RT * i * (i + 1) / 2 > bid) --i;
        ti = i; tj = bid - RT * i * (i + 1) / 2;
    } else {
        const int r = bid - ntri;
        ti = (tri ? nt : 0) + r / nc; tj = r % nc;
    }
    const int kstart = ktri ? ti * TM : 0;
    const double* Ag = A + (long)(ti * TM) * ld + kstart;
    const double* Bg = B + (long)(tj * TN) * ld + kstart;
    const int nk = (K - kstart) / KB;
    const double* pa[GA];
    const double* pb[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) pa[g] = Ag + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
#pragma unroll
    for (int g = 0; g < GB; ++g) pb[g] = Bg + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
    double acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.;
    // fragment offsets (doubles) inside a stage: row & 7 == lr & 7 for every A fragment, (4 (j & 1) + lp) for B
    int oa[2], ob[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        oa[t] = (wm * (TM / WM) + lr) * KB + 2 * ((4 * t + lq) ^ (lr & 7));
#pragma unroll
        for (int par = 0; par < 2; ++par) ob[t][par] = (wn * (TN / WN) + par * 4 + lp) * KB + 2 * ((4 * t + lq) ^ ((par * 4 + lp) & 7));
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&As[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pa[g] + (long)kt * KB) : "memory");
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&Bs[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pb[g] + (long)kt * KB) : "memory");
        }
    };
    static_assert(MT == 2 && NJ == 8, "written for 32 x 32 wave tiles");
    d2 fa[2][MT], fb[NJ];
    auto rd_a = [&](int buf, int t, d2* f) {
#pragma unroll
        for (int i = 0; i < MT; ++i) f[i] = *(const d2*)(&As[buf][oa[t] + i * 16 * KB]);
    };
    auto rd_b = [&](int buf, int t, int j) { fb[j] = *(const d2*)(&Bs[buf][ob[t][j & 1] + (j >> 1) * 8 * KB]); };
    // one half-stage: 4 groups of 8 MFMAs (two B column groups each); as soon as a group has issued, its two B
    // fragments are re-filled for the NEXT half (pbuf, pt); the A fragments of the next half go to the other set
    auto half = [&](const d2* a, bool pre, int pbuf, int pt, d2* anext) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = 2 * g, j1 = 2 * g + 1;
            acc[0][j0] = mfma4(a[0].x, fb[j0].x, acc[0][j0]);
            acc[1][j0] = mfma4(a[1].x, fb[j0].x, acc[1][j0]);
            acc[0][j1] = mfma4(a[0].x, fb[j1].x, acc[0][j1]);
            acc[1][j1] = mfma4(a[1].x, fb[j1].x, acc[1][j1]);
            acc[0][j0] = mfma4(a[0].y, fb[j0].y, acc[0][j0]);
            acc[1][j0] = mfma4(a[1].y, fb[j0].y, acc[1][j0]);
            acc[0][j1] = mfma4(a[0].y, fb[j1].y, acc[0][j1]);
            acc[1][j1] = mfma4(a[1].y, fb[j1].y, acc[1][j1]);
            __builtin_amdgcn_sched_barrier(0);
            if (pre) {
                rd_b(pbuf, pt, j0);
                rd_b(pbuf, pt, j1);
                if (g == 0) rd_a(pbuf, pt, anext);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    rd_a(0, 0, fa[0]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) rd_b(0, 0, j);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int b1 = buf == 2 ? 0 : buf + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        if (kt + 2 < nk) issue(kt + 2, b2);
        half(fa[0], true, buf, 1, fa[1]);                       // t = 0, prefetching t = 1 of this stage
        if (kt + 1 < nk) {
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        half(fa[1], kt + 1 < nk, b1, 0, fa[0]);                 // t = 1, prefetching t = 0 of the next stage
        buf = b1;
    }
    __syncthreads();        // the next tile's prologue overwrites stage buffers
    double* Cg = C + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN + wn * (TN / WN) + lp;
    if (beta != 0.) {
        double cv[MT][NJ];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) cv[i][j] = Cg[(long)(i * 16) * ldc + j * 4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j] + beta * cv[i][j];
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j];
    }
    }
}

// (developer experiment, generalised form of k_gemm_nt_sp2; not in the product header)
// ------------------------------------------------------------------------------------
// K5 (product form): C = beta C + alpha A B^T, software-pipelined.
//   * stages of KB = 16 go global -> LDS DIRECTLY (global_load_lds_dwordx4, 1 KiB = 8 rows per wave instruction,
//     no staging registers, no ds_write), two stages ahead, into 3 LDS buffers; rows are unpadded and the 16-byte
//     chunk c of row r sits at position c ^ (r & 7) (the loading lane fetches the global chunk
//     (lane & 7) ^ (lane >> 3)), so the fragment reads of 8 consecutive rows hit 8 different bank groups;
//   * the B fragments of a half-stage are re-filled from LDS for the NEXT half-stage as soon as the 8 MFMAs that
//     use them have issued, the A fragments are double-buffered: LDS latency hides under this wave's own MFMAs;
//   * ONE barrier per stage, placed between its two halves: it publishes stage kt + 1 (which the second half
//     prefetches from) and retires the reads of the buffer stage kt + 2 will land in.
// Same tile maps, same accumulation order per C element as k_gemm_nt<TM, TN, WM, WN, 16> (bit-identical
// results, tools/gemm_insitu.hip); in situ 7.03 vs 7.45 ms over the 63 updates of an N = 8192 fit.
// The loads are inline assembly (M0 = LDS address of the wave's 1-KiB slot) so that the compiler's own
// s_waitcnt bookkeeping does not serialise them against the fragment reads; the waits are explicit.
// ------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_sp(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int tile_skip, int ktri,
    int tile_end = 0) {
    constexpr int KB = 16, NW = WM * WN, NBUF = 3;
    constexpr int MT = TM / WM / 16, NJ = TN / WN / 4;
    constexpr int GA = TM / 8 / NW, GB = TN / 8 / NW;
    static_assert(GA >= 1 && GB >= 1 && GA + GB < 16 && TM % (8 * NW) == 0 && TN % (8 * NW) == 0, "tile too small for the workgroup");
    static_assert(TM == (TM / TN) * TN && (TM / TN == 1 || TM / TN == 2), "tile shape");
    constexpr int WAIT_NEXT = 0x0F70 | (GA + GB);     // s_waitcnt vmcnt(GA + GB): the stage issued last may still be in flight
    constexpr int WAIT_ALL = 0x0F70;                 // s_waitcnt vmcnt(0)
    __shared__ __attribute__((aligned(128))) double As[NBUF][TM * KB];
    __shared__ __attribute__((aligned(128))) double Bs[NBUF][TN * KB];
    const int bz = blockIdx.z;
    C += (long)bz * sC; A += (long)bz * sAB; B += (long)bz * sAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    constexpr int RT = TM / TN;
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot_all = ntri + (nr - (tri ? nt : 0)) * nc;
    const int ntot = tile_end > 0 ? min(tile_end, ntot_all) : ntot_all;
    const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
    for (int bid = blockIdx.x + tile_skip; bid < ntot; bid += gridDim.x) {
    int ti, tj;
    if (bid < ntri) {
        int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
        while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
        while (RT * i * (i + 1) / 2 > bid) --i;
        ti = i; tj = bid - RT * i * (i + 1) / 2;
    } else {
        const int r = bid - ntri;
        ti = (tri ? nt : 0) + r / nc; tj = r % nc;
    }
    const int kstart = ktri ? ti * TM : 0;
    const double* Ag = A + (long)(ti * TM) * ld + kstart;
    const double* Bg = B + (long)(tj * TN) * ld + kstart;
    const int nk = (K - kstart) / KB;
    const double* pa[GA];
    const double* pb[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) pa[g] = Ag + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
#pragma unroll
    for (int g = 0; g < GB; ++g) pb[g] = Bg + (long)(8 * (g * NW + wave) + lrow) * ld + 2 * gchunk;
    double acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.;
    // fragment offsets (doubles) inside a stage: row & 7 == lr & 7 for every A fragment, (4 (j & 1) + lp) for B
    int oa[2], ob[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        oa[t] = (wm * (TM / WM) + lr) * KB + 2 * ((4 * t + lq) ^ (lr & 7));
#pragma unroll
        for (int par = 0; par < 2; ++par) ob[t][par] = (wn * (TN / WN) + par * 4 + lp) * KB + 2 * ((4 * t + lq) ^ ((par * 4 + lp) & 7));
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&As[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pa[g] + (long)kt * KB) : "memory");
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&Bs[buf][8 * (g * NW + wave) * KB]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(pb[g] + (long)kt * KB) : "memory");
        }
    };
    static_assert(MT == 2 && NJ % 2 == 0, "the MFMA order below keeps 2 MT = 4 instructions between the two updates of an accumulator");
    d2 fa[2][MT], fb[NJ];
    auto rd_a = [&](int buf, int t, d2* f) {
#pragma unroll
        for (int i = 0; i < MT; ++i) f[i] = *(const d2*)(&As[buf][oa[t] + i * 16 * KB]);
    };
    auto rd_b = [&](int buf, int t, int j) { fb[j] = *(const d2*)(&Bs[buf][ob[t][j & 1] + (j >> 1) * 8 * KB]); };
    // one half-stage: 4 groups of 8 MFMAs (two B column groups each); as soon as a group has issued, its two B
    // fragments are re-filled for the NEXT half (pbuf, pt); the A fragments of the next half go to the other set
    auto half = [&](const d2* a, bool pre, int pbuf, int pt, d2* anext) {
#pragma unroll
        for (int g = 0; g < NJ / 2; ++g) {
            const int j0 = 2 * g, j1 = 2 * g + 1;
            acc[0][j0] = mfma4(a[0].x, fb[j0].x, acc[0][j0]);
            acc[1][j0] = mfma4(a[1].x, fb[j0].x, acc[1][j0]);
            acc[0][j1] = mfma4(a[0].x, fb[j1].x, acc[0][j1]);
            acc[1][j1] = mfma4(a[1].x, fb[j1].x, acc[1][j1]);
            acc[0][j0] = mfma4(a[0].y, fb[j0].y, acc[0][j0]);
            acc[1][j0] = mfma4(a[1].y, fb[j0].y, acc[1][j0]);
            acc[0][j1] = mfma4(a[0].y, fb[j1].y, acc[0][j1]);
            acc[1][j1] = mfma4(a[1].y, fb[j1].y, acc[1][j1]);
            __builtin_amdgcn_sched_barrier(0);
            if (pre) {
                rd_b(pbuf, pt, j0);
                rd_b(pbuf, pt, j1);
                if (g == 0) rd_a(pbuf, pt, anext);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) __builtin_amdgcn_s_waitcnt(WAIT_NEXT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
    __syncthreads();
    rd_a(0, 0, fa[0]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) rd_b(0, 0, j);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int b1 = buf == 2 ? 0 : buf + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        if (kt + 2 < nk) issue(kt + 2, b2);
        half(fa[0], true, buf, 1, fa[1]);                       // t = 0, prefetching t = 1 of this stage
        if (kt + 1 < nk) {
            if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(WAIT_NEXT); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            __syncthreads();
        }
        half(fa[1], kt + 1 < nk, b1, 0, fa[0]);                 // t = 1, prefetching t = 0 of the next stage
        buf = b1;
    }
    __syncthreads();        // the next tile's prologue overwrites stage buffers
    double* Cg = C + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN + wn * (TN / WN) + lp;
    if (beta != 0.) {
        double cv[MT][NJ];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) cv[i][j] = Cg[(long)(i * 16) * ldc + j * 4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j] + beta * cv[i][j];
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) Cg[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j];
    }
    }
}


// XCD-aligned tile list: workgroup p runs on XCD p % 8; XCD x gets, in row-major order, the tiles whose column
// tile index is congruent to x mod 8 (so it only ever touches 1/8 of the B panels, also in the triangular part)
static std::vector<int2> build_tab(int nr, int nc, int tri, int RT) {
    std::vector<std::vector<int2>> lists(8);
    const int nt = nc / RT;
    for (int ti = 0; ti < nr; ++ti) {
        const int ncols = (tri && ti < nt) ? RT * (ti + 1) : nc;
        for (int tj = 0; tj < ncols; ++tj) lists[tj & 7].push_back(make_int2(ti, tj));
    }
    size_t L = 0;
    for (auto& l : lists) L = std::max(L, l.size());
    std::vector<int2> tab(8 * L, make_int2(-1, -1));
    for (int x = 0; x < 8; ++x)
        for (size_t q = 0; q < lists[x].size(); ++q) tab[q * 8 + x] = lists[x][q];
    return tab;
}

// (developer experiment, not in the product header: measured equal to the standard forms, the shallow updates are HBM-bound)
// ------------------------------------------------------------------------------------
// K5-flat: the same update for SHALLOW depths (K <= 512: the depth-128 / 256 / 512 updates of the recursion), where a
// tile lasts only 8-32 stages and k_gemm_nt's per-tile prologue (two global round trips before the first MFMA)
// and epilogue are a large share of it.  Persistent workgroups walk their tiles as ONE flat stream of stages:
// the two-stages-ahead prefetch runs across tile boundaries (the first stages of the next tile load under the last
// MFMAs of the current one), and the C values of the next tile are fetched when its first stage is issued.  Batched
// problems are folded into the tile id (no blockIdx.z).  Tile maps, stage order and accumulation order per C
// element are those of k_gemm_nt<TM, TN, WM, WN, 16>: bit-identical results.
// ------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_flat(
    double* __restrict__ C, long ldc, long sC, const double* __restrict__ A, const double* __restrict__ B,
    long ld, long sAB, int K, int nr, int nc, int tri, double alpha, double beta, int nbatch) {
    constexpr int KB = 16, NTH = 64 * WM * WN;
    constexpr int MT = TM / WM / 16, NJ = TN / WN / 4;
    constexpr int KS = KB + 2, CPR = KB / 2;
    constexpr int PA = TM * CPR / NTH, PB = TN * CPR / NTH, RSTEP = NTH / CPR;
    static_assert(PA >= 1 && PB >= 1 && MT * NJ <= 8, "small wave tiles only");
    constexpr int RT = TM / TN;
    static_assert(TM == RT * TN && (RT == 1 || RT == 2), "tile shape");
    __shared__ __attribute__((aligned(16))) double As[2][TM * KS];
    __shared__ __attribute__((aligned(16))) double Bs[2][TN * KS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3;
    const int srow = tid / CPR, sch = (tid % CPR) * 2;
    const int nt = nc / RT;
    const int ntri = tri ? RT * nt * (nt + 1) / 2 : 0;
    const int ntot = ntri + (nr - (tri ? nt : 0)) * nc;          // tiles per problem
    const long total = (long)ntot * nbatch;                      // tiles of the launch
    const int nk = K / KB;                                       // even (K is a multiple of 32)
    // my tiles: blockIdx.x, blockIdx.x + gridDim.x, ...
    const long first = blockIdx.x;
    if (first >= total) return;
    const int ntiles = (int)((total - first + gridDim.x - 1) / gridDim.x);
    const int nstage = ntiles * nk;

    struct Tile { const double* a; const double* b; double* c; };
    auto locate = [&](int it) {
        const long id = first + (long)it * gridDim.x;
        const int z = (int)(id / ntot), bid = (int)(id - (long)z * ntot);
        int ti, tj;
        if (bid < ntri) {
            int i = (int)((sqrtf(8.f * bid / RT + 1.f) - 1.f) * 0.5f);
            while (RT * (i + 1) * (i + 2) / 2 <= bid) ++i;
            while (RT * i * (i + 1) / 2 > bid) --i;
            ti = i;
            tj = bid - RT * i * (i + 1) / 2;
        } else {
            const int r = bid - ntri;
            ti = (tri ? nt : 0) + r / nc;
            tj = r % nc;
        }
        Tile t;
        t.a = A + (long)z * sAB + (long)(ti * TM + srow) * ld + sch;
        t.b = B + (long)z * sAB + (long)(tj * TN + srow) * ld + sch;
        t.c = C + (long)z * sC + (long)(ti * TM + wm * (TM / WM) + ((lane >> 2) & 3) * 4 + lq) * ldc + tj * TN +
              wn * (TN / WN) + lp;
        return t;
    };

    d2 ra[2][PA], rb[2][PB];
    double acc[MT][NJ], cv[MT][NJ], cvn[MT][NJ];
    // load stream: stage index sl = (tile il, depth step kl)
    Tile tl = locate(0), tc = tl;
    int kl = 0, il = 0;
    auto gload = [&](auto par) {           // fetch the load stream's current stage, then advance it (clamped at the end)
        constexpr int P = decltype(par)::value;
        const int k1 = kl * KB;
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[P][p] = *(const d2*)(tl.a + (long)(RSTEP * p) * ld + k1);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[P][p] = *(const d2*)(tl.b + (long)(RSTEP * p) * ld + k1);
        if (kl == 0 && beta != 0.) {       // first stage of a tile: its C values ride along
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) cvn[i][j] = tl.c[(long)(i * 16) * ldc + j * 4];
        }
        if (++kl == nk) {
            if (il + 1 < ntiles) { ++il; kl = 0; tl = locate(il); }
            else kl = nk - 1;              // past the end: re-load the last stage (never stored)
        }
    };
    auto lstore = [&](auto par, int buf) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int p = 0; p < PA; ++p) *(d2*)(&As[buf][(srow + RSTEP * p) * KS + sch]) = ra[P][p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *(d2*)(&Bs[buf][(srow + RSTEP * p) * KS + sch]) = rb[P][p];
    };
    auto compute = [&](int cur) {
        d2 fa[MT], fb[NJ];
#pragma unroll
        for (int t = 0; t < KB / 8; ++t) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                fa[i] = *(const d2*)(&As[cur][(wm * (TM / WM) + i * 16 + lr) * KS + 8 * t + 2 * lq]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                fb[j] = *(const d2*)(&Bs[cur][(wn * (TN / WN) + j * 4 + lp) * KS + 8 * t + 2 * lq]);
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
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { acc[i][j] = 0.; cv[i][j] = 0.; cvn[i][j] = 0.; }
    gload(P0{});                            // stage 0 (+ C of tile 0)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) cv[i][j] = cvn[i][j];
    gload(P1{});                            // stage 1
    lstore(P0{}, 0);
    __syncthreads();
    int kc = 0, ic = 0;                     // compute stream
    for (int s = 0; s < nstage; s += 2) {
        gload(P0{});                        // stage s + 2
        compute(0);
        lstore(P1{}, 1);
        __syncthreads();
        gload(P1{});                        // stage s + 3
        compute(1);
        kc += 2;
        if (kc == nk) {                     // tile finished: write it, take over the C values fetched for the next one
            if (beta != 0.) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) tc.c[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j] + beta * cv[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) tc.c[(long)(i * 16) * ldc + j * 4] = alpha * acc[i][j];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) { acc[i][j] = 0.; cv[i][j] = cvn[i][j]; }
            kc = 0;
            if (++ic < ntiles) tc = locate(ic);
        }
        lstore(P0{}, 0);
        __syncthreads();
    }
}


struct Upd { long a, c0, rows, cols, K; };
static void plan(long c0, long c1, long R, std::vector<Upd>& out) {
    if (c1 - c0 <= 128) return;
    long mid = c0 + (c1 - c0) / 2;
    plan(c0, mid, R, out);
    out.push_back({mid, c0, R - mid, c1 - mid, mid - c0});
    plan(mid, c1, R, out);
}
template <int TM, int TN, int WM, int WN, int KB, int AB = 0>
void run(const char* name, double* M, long ld, const std::vector<Upd>& us) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::map<long, double> ms_by_k, fl_by_k; std::map<long, int> cnt;
    double tot = 0, totfl = 0;
    for (auto& u : us) {
        const long nr = u.rows / TM, nc = u.cols / TN;
        constexpr int RT = TM / TN;
        const long nt = nc / RT;
        const long tiles = RT * nt * (nt + 1) / 2 + (nr - nt) * nc;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_gemm_nt<TM, TN, WM, WN, KB, AB>), dim3(tiles), dim3(64 * WM * WN), 0, 0, M + u.a * ld + u.a, ld, 0L,
                               M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double fl = 2.0 * u.K * (0.5 * u.cols * (u.cols + 1) + (double)(u.rows - u.cols) * u.cols);
        ms_by_k[u.K] += best; fl_by_k[u.K] += fl; cnt[u.K]++; tot += best; totfl += fl;
    }
    printf("%-28s", name);
    for (auto& kv : ms_by_k) printf(" K%-4ld %6.3fms %5.1fTF |", kv.first, kv.second, fl_by_k[kv.first] / (kv.second * 1e-3) / 1e12);
    printf(" total %.3f ms %.1f TF\n", tot, totfl / (tot * 1e-3) / 1e12);
}
template <int TM, int TN, int WM, int WN>
void run_dl(const char* name, double* M, long ld, const std::vector<Upd>& us) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::map<long, double> ms_by_k, fl_by_k;
    double tot = 0, totfl = 0;
    for (auto& u : us) {
        const long nr = u.rows / TM, nc = u.cols / TN;
        constexpr int RT = TM / TN;
        const long nt = nc / RT;
        const long tiles = RT * nt * (nt + 1) / 2 + (nr - nt) * nc;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_gemm_nt_dl<TM, TN, WM, WN>), dim3(tiles), dim3(64 * WM * WN), 0, 0, M + u.a * ld + u.a, ld, 0L,
                               M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double fl = 2.0 * u.K * (0.5 * u.cols * (u.cols + 1) + (double)(u.rows - u.cols) * u.cols);
        ms_by_k[u.K] += best; fl_by_k[u.K] += fl; tot += best; totfl += fl;
    }
    printf("%-28s", name);
    for (auto& kv : ms_by_k) printf(" K%-4ld %6.3fms %5.1fTF |", kv.first, kv.second, fl_by_k[kv.first] / (kv.second * 1e-3) / 1e12);
    printf(" total %.3f ms %.1f TF\n", tot, totfl / (tot * 1e-3) / 1e12);
}

// bitwise comparison of the direct-load kernel with the register-staged one on a 1024 x 512 (tri) update
static int check_dl(double* M, long ld) {
    const long rows = 1024, cols = 512, K = 768, n = rows * ld;
    double *C1, *C2; hipMalloc(&C1, n * 8); hipMalloc(&C2, n * 8);
    hipMemcpy(C1, M, n * 8, hipMemcpyDeviceToDevice); hipMemcpy(C2, M, n * 8, hipMemcpyDeviceToDevice);
    const double* A = M + 2048 * ld;
    const long nr = rows / 128, nc = cols / 64, nt = nc / 2, tiles = 2 * nt * (nt + 1) / 2 + (nr - nt) * nc;
    hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3(tiles), dim3(512), 0, 0, C1, ld, 0L, A, A, ld, 0L, (int)K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
    hipLaunchKernelGGL((k_gemm_nt_sp2<128, 64, 4, 2>), dim3(tiles), dim3(512), 0, 0, C2, ld, 0L, A, A, ld, 0L, (int)K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
    std::vector<double> h1(n), h2(n);
    hipMemcpy(h1.data(), C1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, changed = 0; std::vector<double> h0(n); hipMemcpy(h0.data(), M, n * 8, hipMemcpyDeviceToHost);
    for (long i = 0; i < n; ++i) { if (memcmp(&h1[i], &h2[i], 8)) ++bad; if (h1[i] != h0[i]) ++changed; }
    printf("software-pipelined kernel vs register-staged kernel: %ld differing doubles (of %ld updated)\n", bad, changed);
    hipFree(C1); hipFree(C2);
    return bad != 0;
}

template <int TM, int TN, int WM, int WN, int V = 1>
void run_sp(const char* name, double* M, long ld, const std::vector<Upd>& us) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::map<long, double> ms_by_k, fl_by_k;
    double tot = 0, totfl = 0;
    for (auto& u : us) {
        const long nr = u.rows / TM, nc = u.cols / TN;
        constexpr int RT = TM / TN;
        const long nt = nc / RT;
        const long tiles = RT * nt * (nt + 1) / 2 + (nr - nt) * nc;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (V == 1)
                hipLaunchKernelGGL((k_gemm_nt_sp<TM, TN, WM, WN>), dim3(tiles), dim3(64 * WM * WN), 0, 0, M + u.a * ld + u.a, ld, 0L,
                                   M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            else
                hipLaunchKernelGGL((k_gemm_nt_sp2<TM, TN, WM, WN>), dim3(tiles), dim3(64 * WM * WN), 0, 0, M + u.a * ld + u.a, ld, 0L,
                                   M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double fl = 2.0 * u.K * (0.5 * u.cols * (u.cols + 1) + (double)(u.rows - u.cols) * u.cols);
        ms_by_k[u.K] += best; fl_by_k[u.K] += fl; tot += best; totfl += fl;
    }
    printf("%-28s", name);
    for (auto& kv : ms_by_k) printf(" K%-4ld %6.3fms %5.1fTF |", kv.first, kv.second, fl_by_k[kv.first] / (kv.second * 1e-3) / 1e12);
    printf(" total %.3f ms %.1f TF\n", tot, totfl / (tot * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 8192, tall = argc > 2 ? atol(argv[2]) : 2560, ld = N + (argc > 3 ? atol(argv[3]) : 0), R = N + tall;
    std::vector<double> h((size_t)R * ld);
    srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    double* M; hipMalloc(&M, h.size() * 8);
    hipMemcpy(M, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<Upd> us; plan(0, N, R, us);
    if (argc > 4) {
        // core clock while the K = 4096 update runs (argv[4] = 0: real loads, 1: no global loads)
        int* stop; hipHostMalloc(&stop, 4, hipHostMallocMapped); *stop = 0;
        const int maxn = 40000; long long* out; hipMalloc(&out, (2 * maxn + 1) * 8);
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipLaunchKernelGGL(k_clockmon, dim3(1), dim3(64), 0, s2, stop, out, maxn);
        Upd u{4096, 0, R - 4096, 4096, 4096};
        const long nr = u.rows / 128, nc = u.cols / 64, nt = nc / 2, tiles = 2 * nt * (nt + 1) / 2 + (nr - nt) * nc;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int NREP = argc > 5 ? atoi(argv[5]) : 20;
        usleep(3000);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < NREP; ++rep) {
            if (atoi(argv[4]) >= 100000) {   // (mode 10RRCC: output 128*RR x 128*CC... ) generic: argv[4] = 100000 + 100 * (rows / 1024) + (cols / 1024), K = 4096, separate C
                static double* Cb = nullptr; if (!Cb) hipMalloc(&Cb, (size_t)16384 * 16384 * 8);
                const int rr = ((atoi(argv[4]) - 100000) / 100) * 1024, cc = ((atoi(argv[4]) - 100000) % 100) * 1024;
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3((rr / 128) * (cc / 64)), dim3(512), 0, 0, Cb, 16384L, 0L, M, M, ld, 0L, 4096, rr / 128, cc / 64, 0, -1.0, 1.0, 0, 0);
                if (rep == NREP - 1) printf("[%d x %d x 4096: %d tiles] ", rr, cc, (rr / 128) * (cc / 64));
            } else if (atoi(argv[4]) >= 1000) {   // 4096 x 4096 output (rows 4096.. of M), depth K = argv[4] (<= 8192), no triangle
                const int KK = atoi(argv[4]);
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3(32 * 64), dim3(512), 0, 0, M + 4096 * ld, ld, 0L, M, M, ld, 0L, KK, 32, 64, 0, -1.0, 1.0, 0, 0);
            } else if (atoi(argv[4]) == 6) {   // plain square 4096 x 4096 x 4096, no triangle
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3(32 * 64), dim3(512), 0, 0, M + 4096, ld, 0L, M, M, ld, 0L, 4096, 32, 64, 0, -1.0, 1.0, 0, 0);
            } else if (atoi(argv[4]) == 5) {
                static int2* dtab = nullptr; static int ntab = 0;
                if (!dtab) { auto t = build_tab((int)nr, (int)nc, 1, 2); ntab = (int)t.size(); hipMalloc(&dtab, t.size() * sizeof(int2)); hipMemcpy(dtab, t.data(), t.size() * sizeof(int2), hipMemcpyHostToDevice); printf("table: %d entries for %ld tiles\n", ntab, tiles); }
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3(ntab), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0, 0, (const int2*)dtab, ntab);
            } else if (atoi(argv[4]) == 3)
                hipLaunchKernelGGL((k_gemm_nt_dl<128, 64, 2, 4>), dim3(tiles), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            else if (atoi(argv[4]) == 4)
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 2, 4, 16, 0>), dim3(tiles), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            else if (atoi(argv[4]) == 2)
                hipLaunchKernelGGL((k_gemm_sp<128, 64, 4, 2>), dim3(tiles), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            else if (atoi(argv[4]) == 0)
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 0>), dim3(tiles), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
            else
                hipLaunchKernelGGL((k_gemm_nt<128, 64, 4, 2, 16, 1>), dim3(tiles), dim3(512), 0, 0, M + u.a * ld + u.a, ld, 0L, M + u.a * ld + u.c0, M + u.a * ld + u.c0, ld, 0L, (int)u.K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        usleep(3000);
        *stop = 1; hipStreamSynchronize(s2);
        std::vector<long long> ho(2 * maxn + 1); hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost);
        int n = (int)ho[2 * maxn];
        printf("%d x K4096 update: %.3f ms each (%.1f TF; as a square 4096^3: %.1f TF); %d clock samples:\n", NREP, ms / NREP, 2.0 * 4096 * (0.5 * 4096 * 4097 + (double)(u.rows - 4096) * 4096) / (ms / NREP * 1e-3) / 1e12, (atoi(argv[4]) >= 100000 ? 2.0 * 4096. * 1024. * ((atoi(argv[4]) - 100000) / 100) * 1024. * ((atoi(argv[4]) - 100000) % 100) : 2.0 * 4096. * 4096. * (atoi(argv[4]) >= 1000 ? atof(argv[4]) : 4096.)) / (ms / NREP * 1e-3) / 1e12, n);
        double t = 0;
        for (int i = 0; i < n; ++i) {
            t += ho[2 * i + 1] / 100.0;     // us
            if (i % (n / 12 + 1) == 0) printf("  t=%8.0f us  %6.0f MHz\n", t, ho[2 * i] / (ho[2 * i + 1] / 100.0));
        }
        return 0;
    }
    if (argc > 3) {
        printf("ld = %ld: ", ld);
        run<128, 64, 4, 2, 16>("128x64 8w(32x32)", M, ld, us);
        return 0;
    }
    if (check_dl(M, ld)) return 1;
    run<128, 64, 4, 2, 16>("128x64 8w(32x32)", M, ld, us);
    run_dl<128, 64, 4, 2>("128x64 8w(32x32) direct", M, ld, us);
    run_sp<128, 64, 4, 2>("128x64 8w(32x32) sw-pipelined", M, ld, us);
    run_sp<128, 64, 4, 2, 2>("128x64 8w(32x32) sw-pipelined v2", M, ld, us);
    run_dl<128, 64, 2, 4>("128x64 8w(64x16) direct", M, ld, us);
    run_dl<128, 128, 4, 4>("128x128 16w(32x32) direct", M, ld, us);
    run_dl<64, 64, 2, 4>("64x64 8w(32x16) direct", M, ld, us);
    run<64, 64, 2, 4, 16>("64x64 8w(32x16)", M, ld, us);
    return 0;
    run<128, 64, 2, 4, 16>("128x64 8w(64x16)", M, ld, us);
    run<64, 64, 2, 4, 16>("64x64 8w(32x16)", M, ld, us);
    run<64, 64, 2, 2, 16>("64x64 4w(32x32)", M, ld, us);
    run<64, 32, 2, 2, 16>("64x32 4w(32x16)", M, ld, us);
    run<64, 32, 1, 2, 16>("64x32 2w(64x16)", M, ld, us);
    return 0;
}
