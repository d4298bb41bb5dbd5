// Stage-by-stage probe of k_chol_dag (csrc/dag.hpp) on a small SPD matrix: ./dag_probe <n> <dbg level> <grid>
// Polls the stream instead of blocking, so a hung kernel ends in a message and _exit, not in a stuck process.
#include "../cosmogp_amd/csrc/dag.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 2; } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, dbg = argc > 2 ? atoi(argv[2]) : 0, grid = argc > 3 ? atoi(argv[3]) : 2;
    const int nb = n / 128, nr = nb + 1;
    const long ld = n, R = (long)nr * 128;
    std::vector<double> A((size_t)R * ld, 0.);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[(size_t)i * ld + j] = (i == j ? 2.0 : 0.) + 0.5 * std::exp(-0.001 * (i - j) * (i - j));
    for (int j = 0; j < n; ++j) A[(size_t)(R - 1) * ld + j] = std::sin(0.1 * j);          // residual row
    std::vector<int4> t;
    std::vector<long> applied(nb, 0);
    for (long k = 0; k < nb; ++k) {
        for (long i = k + 2; i < nr; ++i) t.push_back(make_int4((int)i, (int)k, 0, 0));
        for (long j = k + 1; j < nb; ++j)
            for (long i = j; i < nr; ++i) {
                if (i == j && j == k + 1) continue;
                t.push_back(make_int4((int)i, (int)j, (int)k, (int)(((k + 1) << 1) | 1)));
            }
    }
    double *dA, *dW, *dL; int4* dT; unsigned* dS; int* dI;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dW, (size_t)nb * 2048 * 8)); CK(hipMalloc(&dL, nb * 8));
    CK(hipMalloc(&dT, (t.size() + 1) * sizeof(int4))); CK(hipMalloc(&dS, (8 + nr * nb) * 4 + 64)); CK(hipMalloc(&dI, 64));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    if (!t.empty()) CK(hipMemcpy(dT, t.data(), t.size() * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, (8 + nr * nb) * 4 + 64)); CK(hipMemset(dI, 0, 64));
    CK(hipFuncSetAttribute((const void*)k_chol_dag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DAG_LDS_BYTES));
    DagParams p;
    p.M = dA; p.ld = ld; p.nb = nb; p.nr = nr; p.tasks = dT; p.ntasks = (int)t.size(); p.ctrl = dS; p.state = dS + 8;
    p.wsW = dW; p.ldp = dL; p.info = dI; p.timeout = 30000000ll; p.dbg = dbg;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("launch n=%d nb=%d nr=%d tasks=%zu dbg=%d grid=%d\n", n, nb, nr, t.size(), dbg, grid); fflush(stdout);
    hipLaunchKernelGGL(k_chol_dag, dim3(grid), dim3(DAG_THREADS), DAG_LDS_BYTES, st, p);
    CK(hipGetLastError());
    auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { printf("stream error %s\n", hipGetErrorString(q)); return 3; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 8.0) {
            unsigned c[8] = {9, 9, 9, 9, 9, 9, 9, 9};
            printf("HUNG after 8 s\n"); fflush(stdout);
            hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            (void)hipMemcpyAsync(c, dS, 32, hipMemcpyDeviceToHost, s2);
            std::vector<unsigned> stt(nr * nb);
            (void)hipMemcpyAsync(stt.data(), dS + 8, nr * nb * 4, hipMemcpyDeviceToHost, s2);
            (void)hipStreamSynchronize(s2);
            printf("ticket %u abort %u code 0x%x leaf stamp %u worker stamp 0x%x exit 0x%x; states:", c[0], c[1], c[2], c[3], c[4], c[5]);
            for (unsigned v : stt) printf(" %u", v);
            printf("\n"); fflush(stdout);
            _exit(4);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    unsigned c[4]; CK(hipMemcpy(c, dS, 16, hipMemcpyDeviceToHost));
    std::vector<unsigned> stt(nr * nb); CK(hipMemcpy(stt.data(), dS + 8, nr * nb * 4, hipMemcpyDeviceToHost));
    printf("done: ticket %u abort %u code 0x%x stamp %u; states:", c[0], c[1], c[2], c[3]);
    for (unsigned v : stt) printf(" %u", v);
    printf("\n");
    if (dbg == 0) {
        std::vector<double> L((size_t)R * ld); CK(hipMemcpy(L.data(), dA, L.size() * 8, hipMemcpyDeviceToHost));
        // check L L^T = A on the lower triangle and the solved residual row
        double err = 0.;
        for (int i = 0; i < n; i += 7)
            for (int j = 0; j <= i; j += 5) {
                double s = 0.;
                for (int k = 0; k <= j; ++k) s += L[(size_t)i * ld + k] * L[(size_t)j * ld + k];
                err = std::fmax(err, std::fabs(s - A[(size_t)i * ld + j]));
            }
        double zerr = 0.;
        for (int i = 0; i < n; i += 3) {
            double s = 0.;
            for (int k = 0; k <= i; ++k) s += L[(size_t)i * ld + k] * L[(size_t)(R - 1) * ld + k];
            zerr = std::fmax(zerr, std::fabs(s - A[(size_t)(R - 1) * ld + i]));
        }
        printf("max |L L^T - A| = %.3e, max |L z - r| = %.3e\n", err, zerr);
    }
    return 0;
}
