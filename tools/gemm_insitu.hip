// Developer harness: replay the trailing updates of one N x N (+ tall rows) recursive factorisation with the
// fit's own pointer layout and time them per update depth K, for several tile forms.
//   usage: gemm_insitu [N=8192] [tall=2560]
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
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
    run<64, 64, 1, 4, 16>("64x64 4w(64x16)", M, ld, us);
    run<128, 64, 4, 2, 16>("128x64 8w(32x32)", M, ld, us);
    run<128, 64, 2, 4, 16>("128x64 8w(64x16)", M, ld, us);
    run<64, 64, 2, 4, 16>("64x64 8w(32x16)", M, ld, us);
    run<64, 64, 2, 2, 16>("64x64 4w(32x32)", M, ld, us);
    run<64, 32, 2, 2, 16>("64x32 4w(32x16)", M, ld, us);
    run<64, 32, 1, 2, 16>("64x32 2w(64x16)", M, ld, us);
    return 0;
}
