// LDS unaligned-read microbenchmark for gfx950: correctness and cost of ds_read_b32/b64/b96/b128 at byte addresses
// base + stride * lane (stride 1 = 64 consecutive byte positions, the match finder's access pattern), against the
// aligned-dwords + v_alignbyte form.  build: hipcc --offload-arch=gfx950 -O3 -w -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

template <int W>   // bytes per lane: 4, 8, 12, 16
__global__ void k(uint32_t* out, unsigned long long* cyc, int iters, int stride, int misalign, uint32_t* check) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[40960];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = uint8_t(i * 7 + (i >> 8));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t addr = uint32_t(reinterpret_cast<uintptr_t>(lds)) + misalign + stride * lane + (threadIdx.x >> 6) * 8192;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (W == 4) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= v; }
            if (W == 8) { uint64_t v; asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= uint32_t(v) ^ uint32_t(v >> 32); }
            if (W == 12) { uint32_t __attribute__((ext_vector_type(3))) v; asm volatile("ds_read_b96 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= v.x ^ v.y ^ v.z; }
            if (W == 16) { uint4 v; asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(u * 64)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
        addr ^= (acc & 0);   // keep the loop honest
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[0], t1 - t0);
    // correctness: one read compared byte for byte with the LDS contents
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        uint32_t got[4] = {0, 0, 0, 0};
        const uint32_t a = uint32_t(reinterpret_cast<uintptr_t>(lds)) + misalign + stride * lane;
        if (W == 4) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got[0]) : "v"(a));
        if (W == 8) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); got[0] = uint32_t(v); got[1] = uint32_t(v >> 32); }
        if (W == 12) { uint32_t __attribute__((ext_vector_type(3))) v; asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); got[0] = v.x; got[1] = v.y; got[2] = v.z; }
        if (W == 16) { uint4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); got[0] = v.x; got[1] = v.y; got[2] = v.z; got[3] = v.w; }
        uint32_t bad = 0;
        for (int b = 0; b < W; b++) {
            const int i = misalign + stride * lane + b;
            if (uint8_t(got[b >> 2] >> (8 * (b & 3))) != uint8_t(i * 7 + (i >> 8))) bad = 1;
        }
        if (bad) atomicAdd(check, 1u);
    }
}

template <int W>
void run(const char* name) {
    uint32_t* d; unsigned long long* c; uint32_t* chk;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&c, 64); (void)hipMalloc(&chk, 4);
    const int iters = 2000;
    for (int stride : {1, 4, 8, 16, 3}) for (int mis : {0, 1, 2}) for (int wps : {1, 3}) {
        if (stride != 1 && (mis == 2 || wps == 1)) continue;
        (void)hipMemset(c, 0, 16); (void)hipMemset(chk, 0, 4);
        hipLaunchKernelGGL((k<W>), dim3(256 * wps), dim3(256), 0, 0, d, c, iters, stride, mis, chk);
        (void)hipDeviceSynchronize();
        unsigned long long h; uint32_t bad;
        (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost);
        printf("%-14s lane stride %2d B, base misaligned by %d, %d waves/SIMD: %6.1f cycles per wave-read (dependent), %6.1f per CU-read  wrong lanes: %u\n", name, stride, mis, wps,
               double(h) / (iters * 8.0), double(h) / (iters * 8.0 * 4 * wps), bad);
    }
    (void)hipFree(d); (void)hipFree(c); (void)hipFree(chk);
}

int main() {
    run<4>("ds_read_b32");
    run<8>("ds_read_b64");
    run<12>("ds_read_b96");
    run<16>("ds_read_b128");
    return 0;
}
