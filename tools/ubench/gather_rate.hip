// Gather-rate microbenchmark for gfx950: how many lane-addresses per cycle a CU's vector memory path takes, for 1-byte loads, aligned
// and byte-misaligned 4-byte loads, and 4-byte-aligned 16-byte loads; addresses random per lane inside a 1 MiB array per workgroup — 256 MiB in all, 32 MiB per
// XCD: the lines come from the Infinity Cache / HBM, not from the 4 MiB L2 (the regime of the general decoder's phases J and S).  build: hipcc --offload-arch=gfx950 -O3 -w -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>   // 0: u8  1: u32 aligned  2: u32 at odd address  3: 16 B at 4-byte alignment  4: 16 B aligned  5: u8 x 4 consecutive
__global__ __launch_bounds__(1024) void k(const uint8_t* __restrict__ base, uint32_t* out, unsigned long long* cyc, int iters) {
    const uint8_t* b = base + size_t(blockIdx.x) * (1u << 20);
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint32_t a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { x = x * 1664525u + 1013904223u; a[u] = (x >> 9) & ((1u << 20) - 64); }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) acc += b[a[u]];
            if (MODE == 1) { uint32_t v; __builtin_memcpy(&v, b + (a[u] & ~3u), 4); acc += v; }
            if (MODE == 2) { uint32_t v; __builtin_memcpy(&v, b + (a[u] | 1u), 4); acc += v; }
            if (MODE == 3) { uint4 v; __builtin_memcpy(&v, b + ((a[u] & ~15u) | 4u), 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (MODE == 4) { uint4 v; __builtin_memcpy(&v, b + (a[u] & ~15u), 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (MODE == 5) { acc += b[a[u]] + b[a[u] + 1] + b[a[u] + 2] + b[a[u] + 3]; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[0], t1 - t0);
}
template <int MODE> void run(const char* name, int addrs_per_load) {
    uint8_t* d; uint32_t* o; unsigned long long* c;
    (void)hipMalloc(&d, size_t(256) << 20); (void)hipMemset(d, 1, size_t(256) << 20); (void)hipMalloc(&o, 256 * 1024 * 4); (void)hipMalloc(&c, 64);
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, o, c, 5);
    (void)hipMemset(c, 0, 8);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, o, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double loads = double(iters) * 8 * 1024;   // per CU (one workgroup per CU)
    printf("%-34s %6.2f cycles per lane-load per CU  (%5.2f per byte fetched usefully)\n", name, double(h) / loads, double(h) / loads / addrs_per_load);
    (void)hipFree(d); (void)hipFree(o); (void)hipFree(c);
}
int main() {
    run<0>("1-byte loads", 1);
    run<5>("4 x 1-byte loads, consecutive", 4);
    run<1>("4-byte loads, aligned", 4);
    run<2>("4-byte loads, odd address", 4);
    run<4>("16-byte loads, aligned", 16);
    run<3>("16-byte loads, 4-byte aligned", 16);
    return 0;
}
