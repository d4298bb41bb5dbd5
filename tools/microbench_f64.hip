// Developer microbenchmark (not part of the library): what does gfx950 really sustain in fp64?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_f64.hip -o tools/microbench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double* out, unsigned long long* stamps, int iters) {
    d4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    d4 s = c[0];
    for (int i = 1; i < NACC; ++i) s += c[i];
    if (s[0] == 12345.678) out[0] = s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, unsigned long long* stamps, int iters) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = 0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_vfma(double* out, unsigned long long* stamps, int iters) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_fma(c[k], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, double flop_per_wave_iter, int ninst_per_iter, int blocks, int iters) {
    double* out; unsigned long long* st;
    hipMalloc(&out, 64); hipMalloc(&st, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, out, st, iters / 10);
    hipEventRecord(e0);
    launch(blocks, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    double tf = flop_per_wave_iter * 4.0 * blocks * iters / (ms * 1e-3) / 1e12;
    double clk_mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("%-34s blocks %4d  %8.2f TF/s  %7.2f cyc/inst(one wave view)  clock %.0f MHz  %.3f ms\n", name, blocks, tf,
           (double)h[0] / iters / ninst_per_iter, clk_mhz, ms);
    hipFree(out); hipFree(st);
}

#define L16(N) [](int b, double* o, unsigned long long* s, int it) { hipLaunchKernelGGL(k_mfma16<N>, dim3(b), dim3(256), 0, 0, o, s, it); }
#define L4(N) [](int b, double* o, unsigned long long* s, int it) { hipLaunchKernelGGL(k_mfma4<N>, dim3(b), dim3(256), 0, 0, o, s, it); }
#define LV(N) [](int b, double* o, unsigned long long* s, int it) { hipLaunchKernelGGL(k_vfma<N>, dim3(b), dim3(256), 0, 0, o, s, it); }

int main() {
    const int it = 20000;
    for (int blocks : {256, 512, 1024}) {
        run("mfma_f64_16x16x4 acc=1", L16(1), 2048. * 1, 1, blocks, it);
        run("mfma_f64_16x16x4 acc=2", L16(2), 2048. * 2, 2, blocks, it);
        run("mfma_f64_16x16x4 acc=4", L16(4), 2048. * 4, 4, blocks, it);
        run("mfma_f64_16x16x4 acc=8", L16(8), 2048. * 8, 8, blocks, it);
    }
    run("mfma_f64_16x16x4 acc=4 (1 block)", L16(4), 2048. * 4, 4, 1, it);
    for (int blocks : {256, 512}) {
        run("mfma_f64_4x4x4_4b acc=4", L4(4), 512. * 4, 4, blocks, it);
        run("mfma_f64_4x4x4_4b acc=8", L4(8), 512. * 8, 8, blocks, it);
    }
    for (int blocks : {256, 512, 1024, 2048}) {
        run("v_fma_f64 acc=8", LV(8), 128. * 8, 8, blocks, it * 4);
        run("v_fma_f64 acc=16", LV(16), 128. * 16, 16, blocks, it * 4);
    }
    run("v_fma_f64 acc=8 (1 block)", LV(8), 128. * 8, 8, 1, it * 4);
    return 0;
}
