// Developer probe: dependent-chain latency of fp64 VALU ops on one wave (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int NACC>
__global__ void lat(double* out, unsigned long long* st, int iters, double seed) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = seed + i;
    double a = 1.0000001, b = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (OP == 0) c[k] = __builtin_fma(c[k], a, b);
            if (OP == 1) c[k] = c[k] * a;
            if (OP == 2) c[k] = __builtin_amdgcn_rsq(c[k]) + 1.0;   // rsq + add
            if (OP == 3) { int lo = __double2loint(c[k]), hi = __double2hiint(c[k]);
                           lo = __builtin_amdgcn_readlane(lo, 3); hi = __builtin_amdgcn_readlane(hi, 3);
                           c[k] = __builtin_fma(__hiloint2double(hi, lo), a, b); }   // readlane + fma
            if (OP == 4) c[k] = c[k] + a;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += c[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) st[0] = t1 - t0;
}
template <int OP, int NACC> void run(const char* n) {
    double* out; unsigned long long* st; hipMalloc(&out, 512); hipMalloc(&st, 64);
    const int it = 20000;
    hipLaunchKernelGGL((lat<OP, NACC>), dim3(1), dim3(64), 0, 0, out, st, it, 1.5);
    unsigned long long h; hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.1f cycles per op\n", n, NACC, (double)h / it / NACC);
}
int main() {
    run<0, 1>("v_fma_f64"); run<0, 2>("v_fma_f64"); run<0, 4>("v_fma_f64"); run<0, 8>("v_fma_f64");
    run<1, 1>("v_mul_f64"); run<4, 1>("v_add_f64");
    run<2, 1>("v_rsq_f64 + v_add_f64"); run<2, 4>("v_rsq_f64 + v_add_f64");
    run<3, 1>("2x v_readlane + v_fma_f64"); run<3, 4>("2x v_readlane + v_fma_f64");
    return 0;
}
