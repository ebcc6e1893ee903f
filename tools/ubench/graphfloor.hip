// graphfloor.hip -- what a replayed step costs before it computes anything (development tool): a hipGraph of k trivial kernel
// nodes with / without the two copy nodes of rf_env_step's replayed graph, launched and synchronised like the step
// (hipGraphLaunch + hipStreamSynchronize), against the same work enqueued call by call and against a stream that is
// polled (hipStreamQuery) instead of blocked on.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void tiny(int *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d; uint8_t *h;
    CHECK(hipMalloc(&d, 1 << 20));
    CHECK(hipMemset(d, 0, 1 << 20));
    CHECK(hipHostMalloc((void **)&h, 4096, hipHostMallocDefault));
    const int reps = 2000;
    for (int copies = 0; copies <= 2; ++copies)
        for (int k : {1, 2, 4, 6}) {
            for (int mode = 0; mode < 3; ++mode) { // 0: graph + blocking sync, 1: graph + polling, 2: eager + blocking sync
                hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
                auto enqueue = [&]() {
                    if (copies >= 1) CHECK(hipMemcpyAsync(d, h, 12, hipMemcpyHostToDevice, s));
                    for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(i == 2 ? 256 : 1), dim3(256), 0, s, d + 1024);
                    if (copies >= 2) CHECK(hipMemcpyAsync(h + 64, d + 64, 29, hipMemcpyDeviceToHost, s));
                };
                if (mode < 2) {
                    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    enqueue();
                    CHECK(hipStreamEndCapture(s, &g));
                    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                }
                double t0 = 0;
                for (int r = -50; r < reps; ++r) {
                    if (r == 0) t0 = now_us();
                    if (mode < 2) CHECK(hipGraphLaunch(ge, s)); else enqueue();
                    if (mode == 1) { while (hipStreamQuery(s) == hipErrorNotReady) {} }
                    else CHECK(hipStreamSynchronize(s));
                }
                const double us = (now_us() - t0) / reps;
                printf("copies %d kernels %d %s: %.2f us per step\n", copies, k, mode == 0 ? "graph+sync " : mode == 1 ? "graph+poll " : "eager+sync ", us);
                if (ge) CHECK(hipGraphExecDestroy(ge));
                if (g) CHECK(hipGraphDestroy(g));
            }
        }
    return 0;
}
