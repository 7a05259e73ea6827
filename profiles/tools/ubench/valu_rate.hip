// valu_rate.hip -- issue rate of the integer VALU instructions the Myers block step is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed * 3u + threadIdx.x;
    uint64_t w[4] = {seed, seed + 1, seed + 2, seed + 3}, w2 = seed * 77ull; const uint64_t msk = __ballot(threadIdx.x & 1);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 4) asm volatile("v_alignbit_b32 %0, %1, %0, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 7) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 8) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_bfe_u32 %0, %0, 3, 1" : "+v"(a[i]));
            if (OP == 13) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
            if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
            if (OP == 15) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 16) asm volatile("v_cmp_gt_u32_e64 %2, %1, %0\n\tv_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
            if (OP == 17) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[i & 3]));
            if (OP == 18) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 19) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 23) asm volatile("v_xor_b32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 24) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(c));
            if (OP == 25) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i & 3]) : "v"(w2));
            if (OP == 26) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == 27) asm volatile("v_xor_b32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == 28) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)w[i & 3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;       /* 8 blocks of 4 waves per CU = 8 waves per SIMD */
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * ITERS * 8;     /* wave instructions */
    const double per_simd_per_s = insts / 1024 / (ms * 1e-3);
    printf("%-16s %8.3f ms  %6.2f T lane-ops/s  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, insts * 64 / (ms * 1e-3) / 1e12, 2.4e9 / per_simd_per_s);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<1>("v_add_u32", d); run<2>("v_bfi_b32", d); run<3>("v_and_or_b32", d); run<4>("v_alignbit_b32", d);
    run<5>("v_lshl_or_b32", d); run<6>("v_add_co_u32", d); run<7>("v_addc_co_u32", d); run<8>("v_fma_f32", d); run<9>("v_cndmask_b32", d);
    run<10>("v_or3_b32", d); run<11>("v_mov_dpp wshr1", d); run<12>("v_bfe_u32", d); run<13>("v_not_b32", d);
    run<14>("cndmask_e64 sgpr", d); run<15>("cmp+cndmask vcc(2)", d); run<16>("cmp+cndmask sgpr(2)", d); run<17>("v_lshlrev_b64", d); run<18>("v_lshrrev_b32", d);
    run<19>("v_and_b32", d); run<20>("v_xad_u32", d); run<21>("v_add3_u32", d); run<22>("v_mov_b32", d); run<23>("v_xor_b32_e64", d); run<24>("ds_bpermute+wait", d);
    run<25>("v_lshl_add_u64", d); run<26>("v_mov_dpp row_shr", d); run<27>("v_xor_dpp wshr", d); run<28>("v_pk_add_u16", d);
    return 0;
}
