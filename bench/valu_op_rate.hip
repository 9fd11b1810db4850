// Issue rate of the candidate instructions for collecting the signs of the MFMA results (mx_tile) on gfx950: which of them run
// at the 2-cycle rate of v_fma_f32 (wave64), which at 4?  8 independent chains per wave, 4 waves per SIMD, 1 block per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define OP8(S) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc", "s20", "s21")
#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define S_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define S_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define S_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define S_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define S_MAX(i) "v_max_f32_e32 %" #i ", %8, %" #i "\n"
#define S_OR(i) "v_or_b32_e32 %" #i ", %8, %" #i "\n"
#define S_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
#define S_ADD(i) "v_add_u32_e32 %" #i ", %8, %" #i "\n"
#define S_CNDMASK(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n"
#define S_CMP(i) "v_cmp_lt_f32_e32 vcc, %8, %" #i "\n"
#define S_CMPADDC(i) "v_cmp_lt_f32_e32 vcc, %8, %" #i "\n v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define S_PKFMA(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define S_CVTPK(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
#define S_MOV(i) "v_mov_b32_e32 %" #i ", %8\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 7\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define S_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define S_CND64(i) "v_cndmask_b32_e64 %" #i ", %8, %" #i ", s[20:21]\n"
#define S_CNDFMA(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_CMPCND(i) "v_cmp_lt_f32_e32 vcc, %8, %" #i "\n v_cndmask_b32_e32 %" #i ", %9, %" #i ", vcc\n"
#define S_MIN(i) "v_min_f32_e32 %" #i ", %8, %" #i "\n"
#define S_MUL(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n"
#define S_ADDF(i) "v_add_f32_e32 %" #i ", %8, %" #i "\n"
#define S_XOR(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n"
#define S_AND(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n"
#define S_SUB(i) "v_sub_u32_e32 %" #i ", %8, %" #i "\n"
#define S_LSHR(i) "v_lshrrev_b32_e32 %" #i ", 1, %" #i "\n"
#define S_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
template <int M> __global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, int iters, uint32_t seed) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i + seed;
    uint32_t a = 0x3F800001u + seed, b = 0x0F0F0F0Fu + seed;
    asm volatile("" : "+v"(a), "+v"(b));
    asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (M == 0) OP8(S_FMA); if (M == 1) OP8(S_ALIGN); if (M == 2) OP8(S_LSHLOR); if (M == 3) OP8(S_ANDOR); if (M == 4) OP8(S_BFI);
        if (M == 5) OP8(S_PERM); if (M == 6) OP8(S_MAX3); if (M == 7) OP8(S_MAX); if (M == 8) OP8(S_OR); if (M == 9) OP8(S_LSHL);
        if (M == 10) OP8(S_ADD); if (M == 11) OP8(S_CNDMASK); if (M == 12) OP8(S_CMP); if (M == 13) OP8(S_CMPADDC); if (M == 14) OP8(S_PKFMA);
        if (M == 15) OP8(S_CVTPK); if (M == 16) OP8(S_MOV); if (M == 17) OP8(S_BFE); if (M == 18) OP8(S_LSHLADD); if (M == 19) OP8(S_ADD3); if (M == 20) OP8(S_OR3);
        if (M == 21) OP8(S_CND64); if (M == 22) OP8(S_CNDFMA); if (M == 23) OP8(S_CMPCND); if (M == 24) OP8(S_MIN); if (M == 25) OP8(S_MUL); if (M == 26) OP8(S_ADDF);
        if (M == 27) OP8(S_XOR); if (M == 28) OP8(S_AND); if (M == 29) OP8(S_SUB); if (M == 30) OP8(S_LSHR);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t t = 0;
    for (int i = 0; i < 8; ++i) t += x[i];
    if (t == 123u) out[0] = t;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}
template <int M> static void run(const char* name, uint32_t* out, unsigned long long* cyc, int per) {
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(1024), 0, 0, out, cyc, 10, 7u);
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(cyc, 0, 8));
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 7u);
    CHK(hipDeviceSynchronize());
    unsigned long long c; CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-28s %5.2f SIMD-cycles per instruction (4 waves per SIMD)\n", name, (double)c / (blocks * 16.0) / ((double)iters * 8 * per) / 4.0);
}
int main() {
    uint32_t* out; unsigned long long* cyc;
    CHK(hipMalloc((void**)&out, 4096)); CHK(hipMalloc((void**)&cyc, 8));
    run<0>("v_fma_f32", out, cyc, 1); run<1>("v_alignbit_b32", out, cyc, 1); run<2>("v_lshl_or_b32", out, cyc, 1); run<3>("v_and_or_b32", out, cyc, 1);
    run<4>("v_bfi_b32", out, cyc, 1); run<5>("v_perm_b32", out, cyc, 1); run<6>("v_max3_f32", out, cyc, 1); run<7>("v_max_f32", out, cyc, 1);
    run<8>("v_or_b32", out, cyc, 1); run<9>("v_lshlrev_b32", out, cyc, 1); run<10>("v_add_u32", out, cyc, 1); run<11>("v_cndmask_b32", out, cyc, 1);
    run<12>("v_cmp_lt_f32 (vcc)", out, cyc, 1); run<13>("v_cmp_lt_f32 + v_addc_co_u32", out, cyc, 2); run<14>("v_pk_mul_f16", out, cyc, 1);
    run<15>("v_cvt_pkrtz_f16_f32", out, cyc, 1); run<16>("v_mov_b32", out, cyc, 1); run<17>("v_bfe_u32", out, cyc, 1); run<18>("v_lshl_add_u32", out, cyc, 1);
    run<19>("v_add3_u32", out, cyc, 1); run<20>("v_or3_b32", out, cyc, 1);
    run<21>("v_cndmask_b32_e64 (sgpr pair)", out, cyc, 1); run<22>("v_cndmask + v_fma", out, cyc, 2); run<23>("v_cmp + v_cndmask", out, cyc, 2); run<24>("v_min_f32", out, cyc, 1);
    run<25>("v_mul_f32", out, cyc, 1); run<26>("v_add_f32", out, cyc, 1); run<27>("v_xor_b32", out, cyc, 1); run<28>("v_and_b32", out, cyc, 1); run<29>("v_sub_u32", out, cyc, 1); run<30>("v_lshrrev_b32", out, cyc, 1);
    return 0;
}
