// Developer harness: time k_gemm_nt variants in isolation (ablation of global loads / LDS reads).
#ifdef XCD2D
#define CGP_GEMM_XCD2D
#endif
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int TM, int TN, int WM, int WN, int KB, int AB>
void run(const char* name, double* M, long ld, int rows, int cols, int K) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int nr = rows / TM, nc = cols / TN;
    double* C = M + (long)K * 1;   // C region starts at column K (A/B use columns [0,K))
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gemm_nt<TM, TN, WM, WN, KB, AB>), dim3(nr * nc), dim3(64 * WM * WN), 0, 0, C, ld, 0L, M,
                           M, ld, 0L, K, nr, nc, 0, -1.0, 1.0, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s tiles %5d  %.3f ms  %.1f TF/s\n", name, nr * nc, best, 2.0 * rows * cols * K / (best * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const long ld = 8192 + (argc > 4 ? atoi(argv[4]) : 0), R = 12288;
    std::vector<double> h((size_t)R * ld);
    srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    double* M; hipMalloc(&M, h.size() * 8);
    hipMemcpy(M, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const int rows = (argc > 1 ? atoi(argv[1]) : 4096), cols = (argc > 2 ? atoi(argv[2]) : 4096), K = (argc > 3 ? atoi(argv[3]) : 4096);
    run<64, 64, 1, 4, 16, 0>("64x64 4w wave 64x16", M, ld, rows, cols, K);
    run<128, 64, 4, 2, 16, 0>("128x64 8w wave 32x32", M, ld, rows, cols, K);
    run<64, 64, 1, 4, 16, 1>("64x64 4w wave 64x16 no-global", M, ld, rows, cols, K);
    run<128, 64, 4, 2, 16, 1>("128x64 8w wave 32x32 no-global", M, ld, rows, cols, K);
    return 0;
}
