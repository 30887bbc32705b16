// Semantics probes for gfx950 (answers used by the match kernel): does v_alignbyte_b32 look at S2[1:0] only?  what does
// v_ffbl_b32 return for 0?  build: hipcc --offload-arch=gfx950 -O3 -w -o op_semantics op_semantics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    const uint32_t hi = 0x77665544u, lo = 0x33221100u;
    uint32_t r;
    for (uint32_t s = 0; s < 8; s++) {
        const uint32_t sh = s | 0x7ffc;   // high bits set: only [1:0] may matter
        asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "v"(sh));
        out[s] = r;
    }
    uint32_t z = 0;
    asm volatile("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(z));
    out[8] = r;
    uint32_t a = 0xfffffff0u, b = 4096u * 2u;
    asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    out[9] = r;
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    uint32_t h[10]; (void)hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    for (int s = 0; s < 8; s++) printf("v_alignbyte_b32(0x77665544, 0x33221100, 0x7ffc | %d) = 0x%08x\n", s, h[s]);
    printf("v_ffbl_b32(0) = 0x%08x\nv_mul_hi_u32(0xfffffff0, 8192) = %u\n", h[8], h[9]);
    return 0;
}
