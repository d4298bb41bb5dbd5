// Ablation of the covariance assembly (K1): which part of the per-element arithmetic costs what, and what the store
// pattern alone reaches.  ./asm_ablate [n]   (lower triangle of an n x n matrix, 1D and 2D kernels)
//   mode 0 full (as the library)   1 no exp (store the argument)   2 exp of a cheap argument (no divide / sqrt)
//   3 stores only                  4 full, fast divide / sqrt (Markstein correction)   5 = 4 with the custom exp
#include "../cosmogp_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 2; } } while (0)

__device__ __forceinline__ double rbf_exp_notab(double x, const double* tab) {
#pragma clang fp contract(off)
    x = x < -1000. ? -1000. : x;
    const double kb = RBF_FMA(x, RBF_INV_L, 0x1.8p+52);
    const double k = kb - 0x1.8p+52;
    double r = RBF_FMA(-k, RBF_L_HI, x);
    r = RBF_FMA(-k, RBF_L_LO, r);
    int64_t kbits;
    __builtin_memcpy(&kbits, &kb, 8);
    const int ki = (int)(uint32_t)kbits;
    const int m = ki >> 6;
    const double th = tab[0] + (double)(ki & 63), tl = tab[1];
    const double r2 = r * r;
    double p = RBF_FMA(r, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7);
    p = RBF_FMA(p, r, 0x1.5555555555555p-5);
    p = RBF_FMA(p, r, 0x1.5555555555555p-3);
    p = RBF_FMA(p, r, 0.5);
    p = RBF_FMA(p, r2, r);
    const double v = RBF_FMA(th, p, tl) + th;
    return RBF_LDEXP(v, m);
}

template <int DIM, int MODE>
__global__ __launch_bounds__(256) void k_ab(double* __restrict__ M, long ld, const double* __restrict__ x, KParams kp) {
    const int col0 = blockIdx.x * 128, row0 = blockIdx.y * 64;
    if (col0 > row0 + 63) return;
    __shared__ double rx[64 * 2];
    __shared__ __attribute__((aligned(16))) double etab[128];
    const double yinv = rbf_arg_const<DIM>(kp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64 * DIM) rx[tid] = x[(long)row0 * DIM + tid];
    rbf_stage_table(etab);
    const int c = col0 + 2 * lane;
    double ux0, uy0 = 0., ux1, uy1 = 0.;
    if (DIM == 1) { d2 t = *(const d2*)(x + c); ux0 = t.x; ux1 = t.y; }
    else { d2 t0 = *(const d2*)(x + 2 * (long)c), t1 = *(const d2*)(x + 2 * (long)c + 2); ux0 = t0.x; uy0 = t0.y; ux1 = t1.x; uy1 = t1.y; }
    __syncthreads();
#pragma unroll 4
    for (int rr = wave; rr < 64; rr += 4) {
        const int i = row0 + rr;
        const double vx = rx[rr * DIM], vy = (DIM == 2) ? rx[rr * DIM + 1] : 0.;
        d2 o;
        {
#pragma clang fp contract(off)
            if (MODE == 0) {
                o.x = kp.amp_train * rbf_exp(rbf_exp_arg<DIM, DIM == 2>(ux0, uy0, vx, vy, kp, yinv), etab);
                o.y = kp.amp_train * rbf_exp(rbf_exp_arg<DIM, DIM == 2>(ux1, uy1, vx, vy, kp, yinv), etab);
            } else if (MODE == 5) {          // rbfmath exp WITHOUT the table lookup (index forced to 0: wrong values, timing only)
                const double a0 = rbf_exp_arg<DIM, DIM == 2>(ux0, uy0, vx, vy, kp, yinv), a1 = rbf_exp_arg<DIM, DIM == 2>(ux1, uy1, vx, vy, kp, yinv);
                o.x = kp.amp_train * rbf_exp(a0 - a0 * 64. * 0.6931471805599453 * RBF_INV_L * 0. , etab + 0 * (int)a0) ;
                o.y = kp.amp_train * rbf_exp_notab(a1, etab);
                o.x = kp.amp_train * rbf_exp_notab(a0, etab);
            } else if (MODE == 4) {
                o.x = kp.amp_train * exp(rbf_exp_arg<DIM, DIM == 2>(ux0, uy0, vx, vy, kp, yinv));
                o.y = kp.amp_train * exp(rbf_exp_arg<DIM, DIM == 2>(ux1, uy1, vx, vy, kp, yinv));
            } else if (MODE == 1) {
                o.x = kp.amp_train * rbf_exp_arg<DIM, DIM == 2>(ux0, uy0, vx, vy, kp, yinv);
                o.y = kp.amp_train * rbf_exp_arg<DIM, DIM == 2>(ux1, uy1, vx, vy, kp, yinv);
            } else if (MODE == 2) {
                const double d0 = ux0 - vx, d1 = ux1 - vx;
                o.x = kp.amp_train * exp(-0.5 * (d0 * d0) * kp.p1);
                o.y = kp.amp_train * exp(-0.5 * (d1 * d1) * kp.p1);
            } else {
                o.x = vx; o.y = ux0;
            }
        }
        *(d2*)(M + (long)i * ld + c) = o;
    }
}

template <int DIM, int MODE>
double run(double* dM, long n, const double* dx, KParams kp, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((unsigned)(n / 128), (unsigned)(n / 64));
    hipLaunchKernelGGL((k_ab<DIM, MODE>), g, dim3(256), 0, 0, dM, n, dx, kp);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_ab<DIM, MODE>), g, dim3(256), 0, 0, dM, n, dx, kp);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}

__global__ __launch_bounds__(256) void k_store_a(double* out, long n2) {           // as cgp_microbench
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    d2 v = {1.0, 2.0};
    for (; i < n2; i += (long)gridDim.x * blockDim.x) ((d2*)out)[i] = v;
}
__global__ __launch_bounds__(256) void k_store_b(double* out, long n2) {           // 4 x 16 B per lane in flight, 4 KB per wave contiguous
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    d2 v = {1.0, 2.0};
    for (long b = w * 256; b + 255 < n2; b += nw * 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ((d2*)out)[b + q * 64 + lane] = v;
    }
}
__global__ __launch_bounds__(256) void k_store_nt(double* out, long n2) {
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    d2 v = {1.0, 2.0};
    for (long b = w * 256; b + 255 < n2; b += nw * 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(v, (d2*)out + b + q * 64 + lane);
    }
}
template <typename F> double time_store(F f, double* d, long bytes, int blocks, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(d, bytes / 16, blocks);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f(d, bytes / 16, blocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 8192;
    double *dM, *dx;
    CK(hipMalloc(&dM, (size_t)1 << 30)); CK(hipMalloc(&dx, n * 2 * 8));
    std::vector<double> hx(n * 2);
    for (long i = 0; i < 2 * n; ++i) hx[i] = 2048. * ((i * 2654435761u) % 100003) / 100003.;
    CK(hipMemcpy(dx, hx.data(), n * 2 * 8, hipMemcpyHostToDevice));
    KParams kp; kp.amp_train = 1.; kp.amp_cross = 1.; kp.p0 = 110. * 110.; kp.p1 = 1e-4; kp.p2 = 0.; kp.p3 = 1. / (110. * 110.); kp.nug2 = 0.; kp.kss_noise = 0.;
    KParams kp2 = kp; kp2.p0 = 1. / (110. * 110.); kp2.p1 = 0.; kp2.p2 = 0.; kp2.p3 = 1. / (110. * 110.);
    const double bytes = 8. * n * (n + 1) / 2;
    printf("n = %ld, lower triangle %.1f MB\n", n, bytes / 1e6);
    const char* names[5] = {"full (rbfmath)", "no exp", "libm exp of cheap argument", "stores only", "rbfmath argument, libm exp"};
    double t;
    t = run<1, 0>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", names[0], t, bytes / t / 1e3);
    t = run<1, 1>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", names[1], t, bytes / t / 1e3);
    t = run<1, 2>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", names[2], t, bytes / t / 1e3);
    t = run<1, 3>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", names[3], t, bytes / t / 1e3);
    t = run<1, 4>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", names[4], t, bytes / t / 1e3);
    t = run<1, 5>(dM, n, dx, kp, 20); printf("1D %-28s %7.1f us  %6.0f GB/s\n", "rbfmath, no table lookup", t, bytes / t / 1e3);
    t = run<2, 0>(dM, n, dx, kp2, 20); printf("2D %-28s %7.1f us  %6.0f GB/s\n", names[0], t, bytes / t / 1e3);
    t = run<2, 1>(dM, n, dx, kp2, 20); printf("2D %-28s %7.1f us  %6.0f GB/s\n", names[1], t, bytes / t / 1e3);
    t = run<2, 3>(dM, n, dx, kp2, 20); printf("2D %-28s %7.1f us  %6.0f GB/s\n", names[3], t, bytes / t / 1e3);
    t = run<2, 4>(dM, n, dx, kp2, 20); printf("2D %-28s %7.1f us  %6.0f GB/s\n", names[4], t, bytes / t / 1e3);
    t = run<2, 5>(dM, n, dx, kp2, 20); printf("2D %-28s %7.1f us  %6.0f GB/s\n", "rbfmath, no table lookup", t, bytes / t / 1e3);
    for (long mb : {268L, 1024L}) {
        const long by = mb << 20;
        for (int blocks : {1024, 2048, 4096, 8192}) {
            double a = time_store([](double* d, long n2, int b) { hipLaunchKernelGGL(k_store_a, dim3(b), dim3(256), 0, 0, d, n2); }, dM, by, blocks, 10);
            double b = time_store([](double* d, long n2, int b) { hipLaunchKernelGGL(k_store_b, dim3(b), dim3(256), 0, 0, d, n2); }, dM, by, blocks, 10);
            double c = time_store([](double* d, long n2, int b) { hipLaunchKernelGGL(k_store_nt, dim3(b), dim3(256), 0, 0, d, n2); }, dM, by, blocks, 10);
            printf("store %4ld MB, %4d blocks: grid-stride 16 B %6.0f GB/s | 4 KB per wave %6.0f GB/s | nt %6.0f GB/s\n", mb, blocks, a, b, c);
        }
    }
    return 0;
}
