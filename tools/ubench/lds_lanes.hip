// LDS cost of the exec pass's access patterns on gfx950: unaligned reads / writes at RANDOM byte addresses (one copy per lane)
// as a function of the number of ACTIVE lanes and of the access width, beside aligned dword / byte accesses at random
// addresses (bank conflicts only).  Reported: cycles of the CU's LDS pipe per wave instruction (12 waves per CU issuing
// back to back) and the latency seen by a wave alone on the CU.
// build: hipcc --offload-arch=gfx950 -O3 -w -Xclang -target-feature -Xclang +unaligned-ds-access -o lds_lanes lds_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Op { R32A, R64U, R128U, W32A, W8, W64U, W128U, R32U, W32U };

template <int OP>
__global__ void k(uint32_t* out, unsigned long long* cyc, int iters, int active, int align_mask) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[36864];
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = uint8_t(i * 7 + (i >> 8));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // a random byte address per lane (fixed over the loop), inside the first 32 KiB
    uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t addr = uint32_t(reinterpret_cast<uintptr_t>(lds)) + ((h & 32767u) & ~uint32_t(align_mask));
    uint32_t acc = 0;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wv = {h, h + 1, h + 2, h + 3};
    const bool on = lane < active;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (on) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (OP == R32A || OP == R32U) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= v; }
                if (OP == R64U) { uint64_t v; asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= uint32_t(v); }
                if (OP == R128U) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= v.x; }
                if (OP == W32A || OP == W32U) asm volatile("ds_write_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(wv.x), "n"(u * 64) : "memory");
                if (OP == W8) asm volatile("ds_write_b8 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(wv.x), "n"(u * 64) : "memory");
                if (OP == W64U) { uint64_t x = (uint64_t(wv.y) << 32) | wv.x; asm volatile("ds_write_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(x), "n"(u * 64) : "memory"); }
                if (OP == W128U) asm volatile("ds_write_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(wv), "n"(u * 64) : "memory");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) atomicMax(&cyc[0], t1 - t0);
}

template <int OP>
void run(const char* name, int align_mask) {
    uint32_t* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 3 * 256 * 4); (void)hipMalloc(&c, 64);
    const int iters = 500;
    for (int active : {64, 32, 16, 8, 1}) {
        double res[2];
        int q = 0;
        for (int mode : {0, 1}) {   // 0: three 256-thread workgroups per CU (12 waves); 1: one 64-thread workgroup per CU (latency)
            (void)hipMemset(c, 0, 16);
            if (mode == 0) hipLaunchKernelGGL((k<OP>), dim3(256 * 3), dim3(256), 0, 0, d, c, iters, active, align_mask);
            else hipLaunchKernelGGL((k<OP>), dim3(256), dim3(64), 0, 0, d, c, iters, active, align_mask);
            (void)hipDeviceSynchronize();
            unsigned long long h;
            (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            res[q++] = double(h) / (iters * 8.0) / (mode == 0 ? 12.0 : 1.0);
        }
        printf("%-34s active lanes %2d: %6.1f cycles of the CU's LDS pipe per wave instruction, %6.1f cycles latency for a lone wave\n", name, active, res[0], res[1]);
    }
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    run<R32A>("ds_read_b32 aligned, random", 3);
    run<R32U>("ds_read_b32 byte address, random", 0);
    run<R64U>("ds_read_b64 byte address, random", 0);
    run<R128U>("ds_read_b128 byte address, random", 0);
    run<R128U>("ds_read_b128 16-aligned, random", 15);
    run<W8>("ds_write_b8 random", 0);
    run<W32A>("ds_write_b32 aligned, random", 3);
    run<W32U>("ds_write_b32 byte address, random", 0);
    run<W64U>("ds_write_b64 byte address, random", 0);
    run<W128U>("ds_write_b128 byte address, random", 0);
    run<W128U>("ds_write_b128 16-aligned, random", 15);
    return 0;
}
