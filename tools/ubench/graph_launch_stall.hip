// Does the HIP runtime stall once after a certain number of graph launches, whatever the graph contains?
// A graph of `nodes` empty kernels (+ one 4-byte host-to-device copy when `withCopy`) is launched `launches` times on
// one stream, each launch followed by a stream synchronisation; launches that take more than 2 ms are printed.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

static void run(int nodes, bool withCopy, int launches) {
    hipStream_t st; hipStreamCreate(&st);
    int* d; hipMalloc(&d, 4);
    int* h; hipHostMalloc(&h, 4); *h = 7;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    if (withCopy) hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, st);
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d);
    if (withCopy) hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    std::vector<double> ms(launches);
    for (int i = 0; i < launches; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    std::vector<double> s = ms; std::sort(s.begin(), s.end());
    printf("%2d kernel nodes%s, %d launches: median %.3f ms; launches over 2 ms:", nodes, withCopy ? " + 2 copies" : "", launches, s[launches / 2]);
    for (int i = 0; i < launches; ++i) if (ms[i] > 2.0) printf(" #%d %.1f ms", i, ms[i]);
    printf("\n");
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(d); hipHostFree(h); hipStreamDestroy(st);
}

int main(int argc, char** argv) {
    const int nodes = argc > 1 ? atoi(argv[1]) : 20;
    run(nodes, false, 600);
    run(nodes, true, 600);
    return 0;
}
