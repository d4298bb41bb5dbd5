// Developer probe: lane maps + dependent latency of v_mfma_f64_4x4x4_4b_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(const double* a, const double* b, double* d) {
    int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int NACC>
__global__ void lat(double* out, unsigned long long* st, int iters) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = 0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += c[i];
    if (s == 1234.5) out[0] = s;
    if (threadIdx.x == 0) st[0] = t1 - t0;
}
int main() {
    double ha[64], hb[64], hd[64];
    double *a, *b, *d; hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    // hypothesis: A: i = l&3, blk = (l>>2)&3, k = l>>4 ; B: j = l&3, blk = (l>>2)&3, k = l>>4 ; D: j = l&3, blk = (l>>2)&3, i = l>>4
    // test by unit impulses: for each (la, lb) pair set a[la]=1, b[lb]=1 and see which d lanes light up
    int ok = 1;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        for (int i = 0; i < 64; ++i) { ha[i] = 0; hb[i] = 0; }
        ha[la] = 1; hb[lb] = 1;
        hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
        int ai = la & 3, ab = (la >> 2) & 3, ak = la >> 4;
        int bj = lb & 3, bb = (lb >> 2) & 3, bk = lb >> 4;
        for (int l = 0; l < 64; ++l) {
            int dj = l & 3, db = (l >> 2) & 3, di = l >> 4;
            double expect = (ab == bb && ak == bk && db == ab && di == ai && dj == bj) ? 1.0 : 0.0;
            if (hd[l] != expect) {
                if (ok) printf("MISMATCH la=%d lb=%d lane=%d got %g expect %g\n", la, lb, l, hd[l], expect);
                ok = 0;
            }
        }
    }
    printf("layout hypothesis (i|j = l&3, blk = (l>>2)&3, k|i = l>>4): %s\n", ok ? "HOLDS" : "FAILS");
    if (!ok) {   // dump raw map for la,lb in first few
        for (int la = 0; la < 64; la += 1) {
            for (int i = 0; i < 64; ++i) { ha[i] = 0; hb[i] = 1; }
            ha[la] = 1;
            hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
            printf("A lane %2d ->D lanes:", la); for (int l = 0; l < 64; ++l) if (hd[l] != 0) printf(" %d", l); printf("\n");
        }
        for (int lb = 0; lb < 64; lb += 1) {
            for (int i = 0; i < 64; ++i) { ha[i] = 1; hb[i] = 0; }
            hb[lb] = 1;
            hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
            printf("B lane %2d ->D lanes:", lb); for (int l = 0; l < 64; ++l) if (hd[l] != 0) printf(" %d", l); printf("\n");
        }
    }
    unsigned long long* st; double* out; hipMalloc(&st, 64); hipMalloc(&out, 64);
    unsigned long long h;
    const int it = 20000;
#define LAT(N) hipLaunchKernelGGL(lat<N>, dim3(1), dim3(64), 0, 0, out, st, it); hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); \
    printf("4x4x4_4b one wave, %d accumulators: %.2f cycles/inst\n", N, (double)h / it / N);
    LAT(1) LAT(2) LAT(4) LAT(8)
    return 0;
}
