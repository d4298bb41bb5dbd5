// Developer microbenchmark: fp64 MFMA ceilings on RANDOM operands (power/DVFS-limited regime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int WHICH>   // 0: 16x16x4, 1: 4x4x4_4b, 2: v_fma
__global__ __launch_bounds__(256) void k(const double* in, double* out, unsigned long long* stamps, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xFFFFF]; b[i] = in[(t * 16 + 8 + i) & 0xFFFFF]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double res = 0;
    if (WHICH == 0) {
        d4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + it) & 7], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) res += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else if (WHICH == 1) {
        double c[32];
        for (int i = 0; i < 32; ++i) c[i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 7], b[(i >> 2) & 7], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) res += c[i];
    } else {
        double c[32];
        for (int i = 0; i < 32; ++i) c[i] = a[i & 7];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i] = __builtin_fma(c[i], a[i & 7], b[(i >> 2) & 7]);
        }
        for (int i = 0; i < 32; ++i) res += c[i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = res;
    if (threadIdx.x == 0 && blockIdx.x == 7) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main() {
    std::vector<double> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 2.0;
    double *in, *out; unsigned long long* st;
    hipMalloc(&in, h.size() * 8); hipMalloc(&out, 8 << 20); hipMalloc(&st, 64);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"mfma_f64_16x16x4 (rand)", "mfma_f64_4x4x4_4b (rand)", "v_fma_f64 (rand)"};
    for (int which = 0; which < 3; ++which)
        for (int blocks : {256, 512, 1024}) {
            int iters = which == 0 ? 40000 : (which == 1 ? 60000 : 100000);
            double fl = which == 0 ? 8 * 2048. : (which == 1 ? 32 * 512. : 32 * 128.);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
                if (which == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
                if (which == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long s[2]; hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
            printf("%-28s blocks %4d: %7.2f TF/s  clock %4.0f MHz  %.1f ms\n", names[which], blocks,
                   fl * 4 * blocks * iters / (ms * 1e-3) / 1e12, (double)s[0] / s[1] * 100., ms);
        }
    return 0;
}
