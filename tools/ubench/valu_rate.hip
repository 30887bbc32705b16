// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for a few integer ops, by waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t s = seed | 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] + s + i;                                    // v_add3 / v_add
                if (OP == 1) a[i] = a[i] ^ (a[(i + 1) & 7]);                          // v_xor
                if (OP == 2) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 1) & 7], s);  // v_alignbyte
                if (OP == 3) a[i] = a[i] * 2654435761u;                               // v_mul_lo_u32
                if (OP == 4) a[i] = __umul24(a[i], 0x9E3779u) + i;     // v_mad_u32_u24 / v_mul_u32_u24
                if (OP == 5) a[i] = __builtin_ctz(a[i] | 0x80000000u) + a[i];         // v_ffbl + add
                if (OP == 6) a[i] = a[i] > s ? a[i] - s : a[i] + i;                   // cmp + cndmask ...
            }
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int OP>
void run(const char* name, int insts_per_elem) {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 1024 * 4 + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = double(iters) * 64 * insts_per_elem;   // per wave
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (%.2f per wave)\n", name, wps, cyc / (insts * wps), cyc / insts);
    }
    hipFree(d);
}

int main() {
    run<0>("v_add", 1);
    run<1>("v_xor", 1);
    run<2>("v_alignbyte", 1);
    run<3>("v_mul_lo_u32", 1);
    run<4>("v_mul_u24+add", 1);
    run<5>("v_ffbl+or+add", 3);
    run<6>("cmp+cndmask+sub/add", 4);
    return 0;
}
