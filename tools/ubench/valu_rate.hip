// VALU issue-rate microbenchmark for gfx950: SIMD cycles per wave64 integer instruction, by waves per SIMD and by
// the number of independent dependency chains inside a wave.  Cycles come from s_memtime (the shader clock) read by
// the kernel itself, so the answer does not depend on a clock guess; the measured clock is printed beside it.
// build: hipcc --offload-arch=gfx950 -O3 -w -o valu_rate valu_rate.hip ; run on the GPU box (output → profiles/r03_valu_rate.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP, int CH>
__global__ void k(uint32_t* out, unsigned long long* cyc, uint32_t seed, int iters) {
    uint32_t a[CH]; uint64_t b[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = a[i] * 0x100000001ull; }
    uint32_t s = seed | 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / CH; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                // inline asm: exactly one instruction each, nothing for the compiler to fold
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 1) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 2) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 5) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 6) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 7) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 8) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 10) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 11) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 13) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 14) asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 15) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 16) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 17) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 19) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 20) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 21) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 22) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 23) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 24) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 25) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 26) asm volatile("v_alignbyte_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 27) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 28) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 29) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 30) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 31) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 32) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(b[i]) : "v"(s));
                if (OP == 33) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b[i]) : "v"(s));
                if (OP == 34) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 35) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 36) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 37) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 38) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 39) asm volatile("v_readlane_b32 s20, %0, 5" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 40) asm volatile("v_readfirstlane_b32 s20, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 41) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 42) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
                if (OP == 43) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]) : "v"(s) : "vcc", "s20", "s21");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    uint32_t x = 0;
    for (int i = 0; i < CH; i++) x ^= a[i] ^ uint32_t(b[i] >> 13);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) { atomicMax(&cyc[0], t1 - t0); atomicMax(&cyc[1], w1 - w0); }   // the slowest wave: the oldest wave of a SIMD is served first
}

template <int OP, int CH>
void run(const char* name, int insts_per_elem) {
    uint32_t* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 8 * 1024 * 4 + 4096);
    (void)hipMalloc(&c, 64);
    const int iters = 4000;
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, c, 12345u, 10);
        (void)hipMemset(c, 0, 16);
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, c, 12345u, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h[2];
        (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        const double insts = double(iters) * 64 * insts_per_elem;   // per wave
        printf("%-26s chains %2d  waves/SIMD %d: %5.2f SIMD-cycles per wave-instruction (%5.2f cycles between a wave's own instructions)  [s_memtime %.0f MHz]\n",
               name, CH, wps, double(h[0]) / (insts * wps), double(h[0]) / insts, double(h[0]) / (double(h[1]) / 100.0));
    }
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    run<0, 1>("v_add_u32 (VOP2)", 1);
    run<0, 2>("v_add_u32 (VOP2)", 1);
    run<0, 4>("v_add_u32 (VOP2)", 1);
    run<17, 1>("v_mul_lo_u32", 1);
    run<17, 2>("v_mul_lo_u32", 1);
    run<0, 8>("v_add_u32 (VOP2)", 1);
    run<1, 8>("v_add_u32_e64 (VOP3 enc)", 1);
    run<2, 8>("v_sub_u32", 1);
    run<3, 8>("v_xor_b32", 1);
    run<4, 8>("v_and_b32", 1);
    run<5, 8>("v_or_b32", 1);
    run<6, 8>("v_lshlrev_b32", 1);
    run<7, 8>("v_lshrrev_b32", 1);
    run<8, 8>("v_lshrrev_b32 (vgpr amt)", 1);
    run<9, 8>("v_min_u32", 1);
    run<10, 8>("v_max_u32", 1);
    run<11, 8>("v_mov_b32", 1);
    run<12, 8>("v_cndmask_b32 (vcc)", 1);
    run<13, 8>("v_cmp_gt_u32 vcc", 1);
    run<14, 8>("v_cmp_eq_u32 s[..] (e64)", 1);
    run<15, 8>("v_mul_u32_u24 (VOP2)", 1);
    run<16, 8>("v_mad_u32_u24", 1);
    run<17, 8>("v_mul_lo_u32", 1);
    run<18, 8>("v_add3_u32", 1);
    run<19, 8>("v_lshl_add_u32", 1);
    run<20, 8>("v_add_lshl_u32", 1);
    run<21, 8>("v_and_or_b32", 1);
    run<22, 8>("v_xad_u32", 1);
    run<23, 8>("v_bfe_u32", 1);
    run<24, 8>("v_bfi_b32", 1);
    run<25, 8>("v_alignbit_b32", 1);
    run<26, 8>("v_alignbyte_b32", 1);
    run<27, 8>("v_perm_b32", 1);
    run<28, 8>("v_ffbl_b32", 1);
    run<29, 8>("v_ffbh_u32", 1);
    run<30, 8>("v_bcnt_u32_b32", 1);
    run<31, 8>("v_mbcnt_lo_u32_b32", 1);
    run<32, 8>("v_lshlrev_b64", 1);
    run<33, 8>("v_lshrrev_b64", 1);
    run<34, 8>("v_add_co_u32+v_addc_co_u32", 2);
    run<35, 8>("v_mov_b32_dpp row_shr:1", 1);
    run<36, 8>("v_max_u32_dpp row_shr:1", 1);
    run<37, 8>("v_add_u32_sdwa BYTE_1", 1);
    run<38, 8>("v_and_b32_sdwa WORD_1", 1);
    run<39, 8>("v_readlane_b32 (->sgpr)", 1);
    run<40, 8>("v_readfirstlane_b32", 1);
    run<41, 8>("v_pk_add_u16", 1);
    run<42, 8>("v_sad_u8", 1);
    run<43, 8>("v_cvt_f32_ubyte0", 1);
    return 0;
}
