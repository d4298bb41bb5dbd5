// Developer harness: phase stamps (s_memtime) of the leaf kernels K3 / K4 on one 128-block.
#define CGP_STAMPS
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    const long ld = 4096, R = 4096;
    std::vector<double> h((size_t)R * ld);
    srand(1);
    for (long i = 0; i < R; ++i) for (long j = 0; j < ld; ++j) h[i * ld + j] = (i == j) ? 40.0 : 0.1 * (rand() / (double)RAND_MAX - 0.5);
    for (long i = 0; i < 128; ++i) for (long j = 0; j < i; ++j) h[j * ld + i] = h[i * ld + j];
    double *M, *ws, *ldp; int* info;
    hipMalloc(&M, h.size() * 8); hipMalloc(&ws, 32 * 2048 * 8); hipMalloc(&ldp, 256); hipMalloc(&info, 4);
    hipMemset(info, 0, 4);
    hipMemcpy(M, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    size_t lds = (size_t)(CGP_NB * CGP_LDA + 8 * 256 + 144) * 8;
    hipFuncSetAttribute((const void*)k_potrf_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_trsm_panel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long st[256];
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(M, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(CGP_K3_THREADS), lds, 0, M, ld, 0L, 0, ws, 0L, ldp, 0L, info, 0);
        hipLaunchKernelGGL(k_trsm_panel<4>, dim3((R - 128) / 64), dim3(256), lds, 0, M, ld, 0L, 0, 128, ws, 0L, 0, (int)(R - 128));
        hipDeviceSynchronize();
    }
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    printf("potrf_diag: load %llu cycles\n", st[1] - st[0]);
    for (int s = 0; s < 8; ++s)
        printf("  s=%d  (a) %6llu  (c) %6llu  (e) %6llu\n", s, st[3 + 4 * s] - st[2 + 4 * s], st[4 + 4 * s] - st[3 + 4 * s],
               (s < 7 ? st[2 + 4 * (s + 1)] : st[40]) - st[4 + 4 * s]);
    printf("  writeback %llu   total %llu cycles\n", st[41] - st[40], st[41] - st[0]);
    printf("trsm: prologue(load S, L, W) %llu\n", st[51] - st[50]);
    for (int jb = 0; jb < 8; ++jb) printf("  jb=%d %llu\n", jb, (jb < 7 ? st[53 + jb] : st[60]) - st[52 + jb]);
    printf("  total %llu cycles\n", st[60] - st[50]);
    int hi; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost); printf("info %d\n", hi);
    return 0;
}
