// graph_launch.hip — does a HIP graph shorten the device-side time of a sequence of 14 dependent kernels (the library's encode + decode step:
// five encode and nine decode launches, six of them 4-9 us stubs) against plain stream launches issued ahead of the device?  (round-5 review, item 5)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/graph_launch tools/ubench/graph_launch.hip      Run on the GPU box.
// The sequence: kernels of ~2 us (one workgroup) and ~60 us (a device-filling grid) in the step's pattern, each depending on the previous one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void small_k(unsigned* p, unsigned n) { if (threadIdx.x == 0) { unsigned v = p[0]; for (unsigned i = 0; i < n; i++) v = v * 1664525u + 1013904223u; p[0] = v; } }
__global__ void big_k(unsigned* p, unsigned n) {
    unsigned v = p[(blockIdx.x * 256 + threadIdx.x) & 4095];
    for (unsigned i = 0; i < n; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x12345u) p[0] = v;
}

int main() {
    unsigned* d; CK(hipMalloc(&d, 4096 * 4)); CK(hipMemset(d, 1, 4096 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // the step's pattern: B = device-filling kernel, s = stub
    const char* pat = "sBBsBsBsBBssBs";   // far, tiles, serialize, layout, gather | header, exit, chain, index1, index2, viol, schedule, exec, finish
    auto enqueue = [&](hipStream_t s) {
        for (const char* c = pat; *c; c++) {
            if (*c == 'B') hipLaunchKernelGGL(big_k, dim3(4096), dim3(256), 0, s, d, 700u);
            else hipLaunchKernelGGL(small_k, dim3(1), dim3(64), 0, s, d, 300u);
        }
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    for (int i = 0; i < 5; i++) enqueue(st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) enqueue(st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms_stream = 0; CK(hipEventElapsedTime(&ms_stream, e0, e1));
    // the same sequence as a graph (captured once, launched reps times)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    enqueue(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms_graph = 0; CK(hipEventElapsedTime(&ms_graph, e0, e1));
    // each kernel alone, back to back (no dependency pattern change: the sum of the kernels' own durations)
    float sum = 0;
    for (const char* c = pat; *c; c++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; i++) { if (*c == 'B') hipLaunchKernelGGL(big_k, dim3(4096), dim3(256), 0, st, d, 700u); else hipLaunchKernelGGL(small_k, dim3(1), dim3(64), 0, st, d, 300u); }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); sum += t / 50;
    }
    printf("14-kernel step (6 device-filling, 8 one-workgroup stubs), %d repetitions, device time per step:\n", reps);
    printf("  stream launches, host running ahead : %.1f us\n", ms_stream / reps * 1e3);
    printf("  one hipGraphLaunch per step         : %.1f us\n", ms_graph / reps * 1e3);
    printf("  sum of the kernels' own back-to-back durations: %.1f us\n", sum * 1e3);
    return 0;
}
