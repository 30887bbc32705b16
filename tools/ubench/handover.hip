// Hand-over latency between two workgroups on gfx950 (round 5, the question behind "more than one CU per general block"): a ring of G
// workgroups passes a token; each hop = the previous workgroup writes a 32 KiB tile with write-through (sc1) 16-byte stores, waits for its
// stores (vmcnt(0)), barrier, sets a flag (relaxed agent-scope store); the next one polls the flag (relaxed agent-scope load), acquire
// fence, reads all 32 KiB back (16-byte loads, one per thread x 2), barrier.  Reported: microseconds per hop, for rings whose workgroups
// sit on ONE XCD (blockIdx.x % 8 equal: workgroup b is observed to run on XCD b % 8) and on different XCDs, with and without the data.
// build: hipcc --offload-arch=gfx950 -O3 -w -o handover handover.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void store16_wt(uint8_t* p, uint4 v) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
// grid = 64 workgroups; only those with (blockIdx.x % stride == 0 && blockIdx.x / stride < G) take part: stride 8 = same XCD, stride 1 = eight different XCDs
template <int DATA>
__global__ __launch_bounds__(1024) void ring(uint8_t* tiles, uint32_t* flags, unsigned long long* out, int G, int stride, int hops, uint32_t* xcc_out) {
    if (blockIdx.x % stride != 0 || int(blockIdx.x / stride) >= G) return;
    const int me = blockIdx.x / stride, tid = threadIdx.x;
    if (tid == 0) xcc_out[me] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;   // XCC_ID
    uint8_t* mine = tiles + size_t(me) * 32768;
    const uint8_t* prev = tiles + size_t((me + G - 1) % G) * 32768;
    uint32_t acc = 0;
    unsigned long long t0 = 0;
    for (int h = 0; h < hops; h++) {
        if (h % G == me) {   // my turn: wait for hop h's token (flag of the previous workgroup == h), then produce hop h + 1
            if (h > 0) {
                if (tid < 64) { while (__hip_atomic_load(&flags[(me + G - 1) % G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != uint32_t(h)) __builtin_amdgcn_s_sleep(1); }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                if (DATA) {
                    uint4 a, b;
                    __builtin_memcpy(&a, prev + tid * 16, 16); __builtin_memcpy(&b, prev + 16384 + tid * 16, 16);
                    acc += a.x ^ b.y;
                }
            } else if (me == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
            if (DATA) {
                store16_wt(mine + tid * 16, uint4{uint32_t(h), acc, 3, 4});
                store16_wt(mine + 16384 + tid * 16, uint4{uint32_t(h), acc, 5, 6});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&flags[me], uint32_t(h + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == (hops - 1) % G && tid == 0) out[1] = __builtin_amdgcn_s_memrealtime();
    if (me == 0 && tid == 0) out[0] = t0;
    if (acc == 0x12345678u) out[2] = acc;
}
template <int DATA> void run(int G, int stride) {
    uint8_t* t; uint32_t *f, *x; unsigned long long* o;
    (void)hipMalloc(&t, 64 * 32768); (void)hipMalloc(&f, 256); (void)hipMalloc(&x, 256); (void)hipMalloc(&o, 64);
    (void)hipMemset(f, 0, 256); (void)hipMemset(o, 0, 64); (void)hipMemset(t, 0, 64 * 32768);
    const int hops = 2000;
    hipLaunchKernelGGL((ring<DATA>), dim3(64), dim3(1024), 0, 0, t, f, o, G, stride, hops, x);
    (void)hipDeviceSynchronize();
    unsigned long long h[3]; uint32_t xc[8];
    (void)hipMemcpy(h, o, 24, hipMemcpyDeviceToHost); (void)hipMemcpy(xc, x, 32, hipMemcpyDeviceToHost);
    printf("G=%d %-14s %-10s %6.2f us per hop   (XCCs:", G, stride == 8 ? "same XCD" : "across XCDs", DATA ? "32 KiB" : "flag only", double(h[1] - h[0]) / 100.0 / (hops - 1));
    for (int i = 0; i < G; i++) printf(" %u", xc[i]);
    printf(")\n");
    (void)hipFree(t); (void)hipFree(f); (void)hipFree(x); (void)hipFree(o);
}
int main() {
    for (int G : {2, 4}) {
        run<0>(G, 8); run<1>(G, 8);
        run<0>(G, 1); run<1>(G, 1);
    }
    return 0;
}
