// Developer probe: does a latency-critical small kernel (4 waves, big LDS, dependent FMA chain) get CU
// resources promptly while a bulk 8-wave 128x128 GEMM (1 WG/CU by VGPRs, 74 KB LDS) fills the chip?
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void chainlike(double* out, int iters) {
    extern __shared__ double sm[];
    double v = threadIdx.x * 1e-3 + 1.0;
    sm[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) v = __builtin_fma(v, 0.999999, 1e-7);   // dependent fp64 chain
    }
    __syncthreads();
    out[threadIdx.x] = v + sm[(threadIdx.x + 1) & 255];
}
int main(int argc, char** argv) {
    const long ld = 8192, R = 8192;
    std::vector<double> h((size_t)R * ld);
    srand(1); for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    double *M, *out; hipMalloc(&M, h.size() * 8); hipMalloc(&out, 4096);
    hipMemcpy(M, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, -1); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, g0, g1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&g0); hipEventCreate(&g1);
    const int lds_small[3] = {40 * 1024, 86 * 1024, 150 * 1024};
    hipFuncSetAttribute((const void*)chainlike, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nchain = (argc > 1 ? atoi(argv[1]) : 40), iters = 1500; const int tdim = (argc > 2 ? atoi(argv[2]) : 4096);   // ~1500 dependent FMAs ~ 48k cycles ~ 20 us
    for (int which = 0; which < 3; ++which) {
        // alone
        hipEventRecord(e0, sa);
        for (int i = 0; i < nchain; ++i) hipLaunchKernelGGL(chainlike, dim3(1), dim3(256), lds_small[which], sa, out, iters);
        hipEventRecord(e1, sa); hipEventSynchronize(e1);
        float alone; hipEventElapsedTime(&alone, e0, e1);
        // beside the bulk GEMM (4096 x 4096 x 4096, 128x128 tiles, 8 waves): ~2.5 ms
        const int rows = tdim, cols = tdim, K = 4096, nr = rows / 128, nc = cols / 128;
        hipEventRecord(g0, sb);
        hipLaunchKernelGGL((k_gemm_nt<128, 128, 2, 4, 16, 0>), dim3(nr * nc), dim3(512), 0, sb, M + K, ld, 0L, M, M, ld, 0L, K, nr, nc, 0, -1.0, 1.0, 0, 0);
        hipEventRecord(g1, sb);
        hipEventRecord(e0, sa);
        for (int i = 0; i < nchain; ++i) hipLaunchKernelGGL(chainlike, dim3(1), dim3(256), lds_small[which], sa, out, iters);
        hipEventRecord(e1, sa); hipEventSynchronize(e1); hipEventSynchronize(g1);
        float beside, gemm; hipEventElapsedTime(&beside, e0, e1); hipEventElapsedTime(&gemm, g0, g1);
        // gemm alone
        hipEventRecord(g0, sb);
        hipLaunchKernelGGL((k_gemm_nt<128, 128, 2, 4, 16, 0>), dim3(nr * nc), dim3(512), 0, sb, M + K, ld, 0L, M, M, ld, 0L, K, nr, nc, 0, -1.0, 1.0, 0, 0);
        hipEventRecord(g1, sb); hipEventSynchronize(g1);
        float galone; hipEventElapsedTime(&galone, g0, g1);
        printf("small kernel LDS %3d KB: %d launches alone %.1f us each; beside bulk GEMM %.1f us each; GEMM %.2f ms (alone %.2f ms)\n",
               lds_small[which] / 1024, nchain, alone * 1e3 / nchain, beside * 1e3 / nchain, gemm, galone);
    }
    return 0;
}
