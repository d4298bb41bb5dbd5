// Developer probe: latency of cross-stream event dependencies (two streams ping-ponging tiny kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
int main() {
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int N = 2000;
    hipEvent_t ev[2 * N]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    // same-stream baseline
    hipEventRecord(t0, s1);
    for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
    hipEventRecord(t1, s1); hipEventSynchronize(t1);
    float ms; hipEventElapsedTime(&ms, t0, t1);
    printf("same stream: %.2f us per kernel\n", ms * 1e3 / (2 * N));
    auto h0 = std::chrono::high_resolution_clock::now();
    hipEventRecord(t0, s1);
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
        hipEventRecord(ev[2 * i], s1);
        hipStreamWaitEvent(s2, ev[2 * i], 0);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d);
        hipEventRecord(ev[2 * i + 1], s2);
        hipStreamWaitEvent(s1, ev[2 * i + 1], 0);
    }
    hipEventRecord(t1, s1);
    auto h1 = std::chrono::high_resolution_clock::now();
    hipEventSynchronize(t1);
    hipEventElapsedTime(&ms, t0, t1);
    printf("ping-pong across 2 streams: %.2f us per hop (host enqueue %.2f us per hop)\n", ms * 1e3 / (2 * N),
           std::chrono::duration<double, std::micro>(h1 - h0).count() / (2 * N));
    // graph capture of the same pattern
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
        hipEventRecord(ev[2 * i], s1);
        hipStreamWaitEvent(s2, ev[2 * i], 0);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d);
        hipEventRecord(ev[2 * i + 1], s2);
        hipStreamWaitEvent(s1, ev[2 * i + 1], 0);
    }
    hipStreamEndCapture(s1, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s1); hipStreamSynchronize(s1);
    hipEventRecord(t0, s1); hipGraphLaunch(ge, s1); hipEventRecord(t1, s1); hipEventSynchronize(t1);
    hipEventElapsedTime(&ms, t0, t1);
    printf("same pattern as a captured graph: %.2f us per hop\n", ms * 1e3 / (2 * N));
    return 0;
}
