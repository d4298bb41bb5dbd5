// Developer harness: replay the trailing updates of one N x N (+ tall rows) recursive factorisation with the
// fit's own pointer layout and time them per update depth K, for several tile forms.
//   usage: gemm_insitu [N=8192] [tall=2560]
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
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
    hipLaunchKernelGGL((k_gemm_nt_dl<128, 64, 4, 2>), dim3(tiles), dim3(512), 0, 0, C2, ld, 0L, A, A, ld, 0L, (int)K, (int)nr, (int)nc, 1, -1.0, 1.0, 0, 0);
    std::vector<double> h1(n), h2(n);
    hipMemcpy(h1.data(), C1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, changed = 0; std::vector<double> h0(n); hipMemcpy(h0.data(), M, n * 8, hipMemcpyDeviceToHost);
    for (long i = 0; i < n; ++i) { if (memcmp(&h1[i], &h2[i], 8)) ++bad; if (h1[i] != h0[i]) ++changed; }
    printf("direct-load kernel vs register-staged kernel: %ld differing doubles (of %ld updated)\n", bad, changed);
    hipFree(C1); hipFree(C2);
    return bad != 0;
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
        const int maxn = 4000; long long* out; hipMalloc(&out, (2 * maxn + 1) * 8);
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipLaunchKernelGGL(k_clockmon, dim3(1), dim3(64), 0, s2, stop, out, maxn);
        Upd u{4096, 0, R - 4096, 4096, 4096};
        const long nr = u.rows / 128, nc = u.cols / 64, nt = nc / 2, tiles = 2 * nt * (nt + 1) / 2 + (nr - nt) * nc;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        usleep(3000);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 20; ++rep) {
            if (atoi(argv[4]) == 0)
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
        printf("20 x K4096 update: %.3f ms each (%.1f TF); %d clock samples:\n", ms / 20, 2.0 * 4096 * (0.5 * 4096 * 4097 + (double)(u.rows - 4096) * 4096) / (ms / 20 * 1e-3) / 1e12, n);
        double t = 0;
        for (int i = 0; i < n; ++i) {
            t += ho[2 * i + 1] / 100.0;     // us
            if (i % (n / 40 + 1) == 0) printf("  t=%8.0f us  %6.0f MHz\n", t, ho[2 * i] / (ho[2 * i + 1] / 100.0));
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
