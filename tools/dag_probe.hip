// Stage-by-stage probe of k_chol_dag (csrc/dag.hpp) on a small SPD matrix: ./dag_probe <n> <dbg level> <grid>
// Polls the stream instead of blocking, so a hung kernel ends in a message and _exit, not in a stuck process.
#include "../cosmogp_amd/csrc/dag.hpp"
#include "../cosmogp_amd/csrc/dag_tasks.hpp"
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
    const int tall = argc > 4 ? atoi(argv[4]) : 1, depth = argc > 5 ? atoi(argv[5]) : 1, window = argc > 6 ? atoi(argv[6]) : depth;
    DagSchedule sch; sch.depth = depth; sch.window = window; sch.workers = grid - 1;
    if (argc > 7) sch.dmax = atoi(argv[7]);
    if (argc > 8) sch.fill = 0.01 * atoi(argv[8]);
    if (argc > 9) sch.lead = atoi(argv[9]);
    if (argc > 10) sch.slope = atoi(argv[10]);
    const int nmat = getenv("DAG_PROBE_NMAT") ? atoi(getenv("DAG_PROBE_NMAT")) : 1;      // objects side by side (same matrix in each)
    const int nb = n / 128, nr = nb + tall;
    const long ld = n, R = (long)nr * 128;
    std::vector<double> A((size_t)R * ld, 0.);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[(size_t)i * ld + j] = (i == j ? 2.0 : 0.) + 0.5 * std::exp(-0.001 * (i - j) * (i - j));
    for (int j = 0; j < n; ++j) A[(size_t)(R - 1) * ld + j] = std::sin(0.1 * j);          // residual row
    std::vector<int4> t;
    sch.workers = getenv("DAG_PROBE_SCHED_WORKERS") ? atoi(getenv("DAG_PROBE_SCHED_WORKERS")) : grid - 1;
    dag_build_tasks_batch(nb, nr, nmat, sch, t, getenv("DAG_PROBE_ORDER") ? atoi(getenv("DAG_PROBE_ORDER")) : 1);
    if (dbg == 9) {
        // update-task benchmark: every lower tile once, depth `depth` (k0 = 0), all dependencies pre-satisfied
        t.clear();
        for (int rep = 0; rep < 4; ++rep)
            for (int i = depth; i < nr; ++i)
                for (int j = depth; j <= i && j < nb; ++j) t.push_back(make_int4(i, j, 0, (depth << 1) | 1));
    }
    double *dA, *dW, *dL; int4* dT; unsigned* dS; int* dI;
    CK(hipMalloc(&dA, A.size() * 8 * nmat)); CK(hipMalloc(&dW, (size_t)nb * 2048 * 8 * nmat)); CK(hipMalloc(&dL, nb * 8 * nmat));
    CK(hipMalloc(&dT, (t.size() + 1) * sizeof(int4))); CK(hipMalloc(&dS, (8 + (size_t)nr * nb * nmat) * 4 + 64)); CK(hipMalloc(&dI, 64 + 4 * nmat));
    for (int b = 0; b < nmat; ++b) CK(hipMemcpy(dA + (size_t)b * A.size(), A.data(), A.size() * 8, hipMemcpyHostToDevice));
    if (!t.empty()) CK(hipMemcpy(dT, t.data(), t.size() * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, (8 + (size_t)nr * nb * nmat) * 4 + 64)); CK(hipMemset(dI, 0, 64 + 4 * nmat));
    if (dbg == 9) { std::vector<unsigned> big(nr * nb, 100000u); CK(hipMemcpy(dS + 8, big.data(), big.size() * 4, hipMemcpyHostToDevice)); }
    CK(hipFuncSetAttribute((const void*)k_chol_dag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DAG_LDS_BYTES));
    DagParams p;
    p.sticky = nullptr;
    p.nmat = dbg == 9 ? 1 : nmat; p.sM = (long)A.size(); p.sW = (long)nb * 2048; p.sLd = nb;
    p.M = dA; p.ld = ld; p.nb = nb; p.nr = nr; p.tasks = dT; p.ntasks = (int)t.size(); p.ctrl = dS; p.state = dS + 8;
    unsigned long long* dP; const size_t nprof = (size_t)nb * 8 + (size_t)grid * 4 + 8;
    CK(hipMalloc(&dP, nprof * 8)); CK(hipMemset(dP, 0, nprof * 8));
    p.wsW = dW; p.ldp = dL; p.info = dI; p.timeout = 30000000ll; p.dbg = dbg; p.prof = dP;
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
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const double wall_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
    unsigned c[4]; CK(hipMemcpy(c, dS, 16, hipMemcpyDeviceToHost));
    if (dbg == 9) {
        std::vector<unsigned long long> pr(nprof); CK(hipMemcpy(pr.data(), dP, nprof * 8, hipMemcpyDeviceToHost));
        double w[4] = {0}; int nw = 0;
        for (int g = 1; g < grid; ++g) {
            const unsigned long long* q = &pr[(size_t)nb * 8 + (size_t)g * 4];
            if (q[3] == 0) continue;
            ++nw; for (int cc = 0; cc < 4; ++cc) w[cc] += q[cc] * (cc < 3 ? 0.01 : 1.);
        }
        const double per = w[2] / w[3], unit = 2.0 * 128 * 128 * 128 * depth;
#ifdef DAG_UPD_STAMPS
        { unsigned long long a[8]; CK(hipMemcpyFromSymbol(a, HIP_SYMBOL(g_upd_acc), sizeof(a)));
          if (a[4]) printf("update task of workgroup 1 (us): loads until the first stage is in LDS %.2f, K loop %.2f, negate + issue stores %.2f, drain of the stores %.2f (%llu tasks)\n",
                           a[0] * 0.01 / a[4], a[1] * 0.01 / a[4], a[2] * 0.01 / a[4], a[3] * 0.01 / a[4], a[4]); }
#endif
        printf("updbench depth %d: %zu tasks, %d workers: fetch+wait %.2f us/task, body+publish %.2f us/task -> %.1f GF/s per CU, %.1f TF/s on %d CUs (host wall %.2f ms)\n",
               depth, t.size(), nw, w[0] / w[3], per, unit / per * 1e-3, unit / (per + w[0] / w[3]) * 1e-6 * nw, nw, wall_ms);
        return 0;
    }
    std::vector<unsigned> stt(nr * nb); CK(hipMemcpy(stt.data(), dS + 8, nr * nb * 4, hipMemcpyDeviceToHost));
    printf("done in %.3f ms (host wall): ticket %u abort %u code 0x%x stamp %u", wall_ms, c[0], c[1], c[2], c[3]);
    if (nr * nb <= 64) { printf("; states:"); for (unsigned v : stt) printf(" %u", v); }
    printf("\n");
    if (dbg == 0) {
        std::vector<unsigned long long> pr(nprof); CK(hipMemcpy(pr.data(), dP, nprof * 8, hipMemcpyDeviceToHost));
        // leaf lane: average time per phase (us) over the steps
        double ph[8] = {0}; const double us = 0.01;
        for (int k = 0; k + 1 < nb; ++k) {
            const unsigned long long* q = &pr[(size_t)k * 8];
            ph[0] += (q[1] - q[0]) * us;        // factor
            ph[1] += (q[2] - q[1]) * us;        // write-back issue + wait for the workers' tiles
            ph[2] += (q[3] - q[2]) * us;        // loads of the step, drain, factor published
            ph[3] += (q[4] - q[3]) * us;        // solve, X published
            ph[4] += (q[5] - q[4]) * us;        // X X^T update into LDS
            ph[6] += (pr[(size_t)(k + 1) * 8] - q[0]) * us;   // whole step
        }
        if (getenv("DAG_PROBE_STEPS")) {
            printf("step durations (us), 8 per row:");
            for (int k = 0; k + 1 < nb; ++k) printf("%s%6.1f", k % 8 == 0 ? "\n  " : " ", (pr[(size_t)(k + 1) * 8] - pr[(size_t)k * 8]) * us);
            printf("\n");
        }
        const double total = (pr[(size_t)(nb - 1) * 8 + 3] - pr[0]) * us;
        printf("leaf lane per step (us): factor %.2f  writeback+wait %.2f  loads+publish %.2f  solve %.2f  update %.2f  step %.2f; chain total %.1f us\n",
               ph[0] / (nb - 1), ph[1] / (nb - 1), ph[2] / (nb - 1), ph[3] / (nb - 1), ph[4] / (nb - 1), ph[6] / (nb - 1), total);
        double w[6] = {0}; int nw = 0;
        for (int g = 1; g < grid; ++g) {
            const unsigned long long* q = &pr[(size_t)nb * 8 + (size_t)g * 4];
            if (q[3] == 0) continue;
            ++nw; for (int c = 0; c < 3; ++c) w[c] += q[c] * us;
            w[3] += (double)(q[3] & 0xfffff); w[4] += (double)((q[3] >> 20) & 0x3fffff) * 16 * us; w[5] += (double)(q[3] >> 42) * 16 * us;
        }
        if (nw) printf("workers (%d active): per worker wait %.1f us (before TRSM %.1f, before one-panel UPD %.1f, before deep UPD %.1f), TRSM %.1f us, UPD %.1f us, %.1f tasks; per task: %.2f us busy\n",
                       nw, w[0] / nw, w[4] / nw, w[5] / nw, (w[0] - w[4] - w[5]) / nw, w[1] / nw, w[2] / nw, w[3] / nw, (w[1] + w[2]) / w[3]);
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
        for (int b = 1; b < nmat; ++b) {
            std::vector<double> Lb((size_t)R * ld); CK(hipMemcpy(Lb.data(), dA + (size_t)b * A.size(), Lb.size() * 8, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (int i = 0; i < R; ++i) for (int j = 0; j <= i && j < n; ++j) diff += Lb[(size_t)i * ld + j] != L[(size_t)i * ld + j];
            if (diff || b == nmat - 1) printf("object %d vs object 0: %zu differing entries\n", b, diff);
        }
    }
    return 0;
}
