// Issue cost of the integer VALU instructions the sketch / hash code is made of, on gfx950: cycles per wave64
// instruction and SIMD with 1, 2 and 4 waves per SIMD (eight independent chains per wave, 4096 instructions per lap).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void __launch_bounds__(1024) k(uint64_t* out, int laps, unsigned long long* cyc) {
    uint64_t a0 = threadIdx.x * 0x9E3779B97F4A7C15ULL + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s = (uint32_t)laps | 3u;
    asm volatile("s_mov_b64 s[20:21], 0x55" : : : "s20", "s21");
    const unsigned long long t0 = clock64();
    for (int l = 0; l < laps; ++l) {
#define EACH(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
        if (OP == 0) {
#define M(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 1) {
#define M(a) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a));
            REP64(EACH(M))
#undef M
        } else if (OP == 2) {
#define M(a) asm volatile("v_lshrrev_b64 %0, 5, %0" : "+v"(a));
            REP64(EACH(M))
#undef M
        } else if (OP == 3) {
#define M(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, 21, %0" : "+v"(a) : "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 4) {
#define M(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 5) {
#define M(a) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 6) {
#define M(a) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a), "v"(a0) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 7) {
#define M(a) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a) : "v"(a0));
            REP64(EACH(M))
#undef M
        } else if (OP == 8) {
#define M(a) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 9) {
#define M(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(uint32_t*)&a) : "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 16) {   // mask in an SGPR pair other than vcc
#define M(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(*(uint32_t*)&a) : "v"(s) : "s20", "s21");
            REP64(EACH(M))
#undef M
        } else if (OP == 17) {   // vcc written by a VALU compare before every group of eight selects
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(*(uint32_t*)&a0), "v"(s) : "vcc");
#define M(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(uint32_t*)&a) : "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 18) {   // the select as a bit-field insert under a VGPR mask
#define M(a) asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s), "v"(*(uint32_t*)&a7));
            REP64(EACH(M))
#undef M
        } else if (OP == 19) {   // compare + select pairs, as compiled code has them
#define M(a) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(uint32_t*)&a) : "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 20) {   // 32-bit maximum (select-free)
#define M(a) asm volatile("v_max_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 10) {
#define M(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 11) {
#define M(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 12) {
#define M(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        } else if (OP == 13) {
#define M(a) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(*(uint32_t*)&a), "v"(s) : "vcc");
            REP64(EACH(M))
#undef M
        } else if (OP == 14) {
#define M(a) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(*(uint32_t*)&a));
            REP64(EACH(M))
#undef M
        } else if (OP == 15) {
#define M(a) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(*(uint32_t*)&a) : "v"(s));
            REP64(EACH(M))
#undef M
        }
    }
    const unsigned long long t1 = clock64();
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char* name, uint64_t* out, unsigned long long* cyc) {
    const int laps = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {   // wps waves on every SIMD of the 256 CUs: blocks of up to 16 waves, one or two per CU
        unsigned long long h = 0;
        const int per_block = wps < 4 ? wps : 4, blocks = 256 * (wps / per_block);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * 4 * per_block), 0, 0, out, laps, cyc);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * 4 * per_block), 0, 0, out, laps, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        // per SIMD: wps waves x laps x 512 instructions in `ms`
        printf("%-16s waves/SIMD %d: %7.3f clock64 ticks per instruction of one wave; kernel %7.1f us -> %6.3f ns per wave-instruction and SIMD\n", name, wps,
               (double)h / (laps * 512.0), ms * 1e3, ms * 1e6 / (laps * 512.0 * wps));
    }
}
int main() {
    uint64_t* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 8);
    run<0>("v_add_u32", out, cyc); run<12>("v_xor_b32", out, cyc); run<1>("v_lshlrev_b64", out, cyc); run<2>("v_lshrrev_b64", out, cyc); run<3>("v_mad_u64_u32", out, cyc);
    run<4>("v_mul_lo_u32", out, cyc); run<11>("v_mul_hi_u32", out, cyc); run<10>("v_mad_u32_u24", out, cyc); run<5>("v_alignbit_b32", out, cyc);
    run<6>("v_cmp_lt_u64", out, cyc); run<13>("v_cmp_lt_u32", out, cyc); run<7>("v_lshl_add_u64", out, cyc); run<8>("v_add_co_u32", out, cyc); run<9>("v_cndmask_b32", out, cyc);
    run<14>("v_bfe_u32", out, cyc); run<15>("v_lshl_or_b32", out, cyc);
    run<16>("cndmask sgpr", out, cyc); run<17>("cndmask vcc set", out, cyc); run<18>("v_bfi_b32", out, cyc); run<19>("cmp+nop+cndmask", out, cyc); run<20>("v_max_u32", out, cyc);
    return 0;
}
