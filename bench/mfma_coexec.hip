// Can the matrix core and the vector ALU of a gfx950 SIMD run side by side?  (round 2 re-check of bench/mfma_overlap.hip, which found
// "times add" for one wave stream of [MFMA, 32 VALU].)  Variants, all with 4 waves per SIMD resident (block = 1024 threads, 1 block/CU):
//   A  every wave: NM x MFMA (independent accumulators) then NV x v_fma_f32 per iteration        (the r01 shape)
//   B  every wave: MFMAs and VALU interleaved 1 : NV/NM in program order
//   C  specialised waves: even waves issue only MFMAs, odd waves only VALU (same totals per SIMD as A)
//   M  MFMA only, V  VALU only (the two terms of the sum)
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_coexec mfma_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int NM, int NV>
__global__ __launch_bounds__(1024) void k(int reps, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(1.f - i * 0.1f); }
    f16v c[4] = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    float m = 1.0001f + threadIdx.x * 1e-9f, ad = 0.5f;
    asm volatile("" : "+v"(m), "+v"(ad));
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 'M' || MODE == 'A' || MODE == 'B' || (MODE == 'C' && (wave & 1) == 0);
    const bool do_v = MODE == 'V' || MODE == 'A' || MODE == 'B' || (MODE == 'C' && (wave & 1) == 1);
    const int nm = MODE == 'C' ? 2 * NM : NM, nv = MODE == 'C' ? 2 * NV : NV; /* C: the same totals per SIMD */
    for (int r = 0; r < reps; ++r) {
        if (MODE == 'B') {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NV / NM; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j * (NV / NM) + i) % 16]) : "v"(m), "v"(ad));
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int j = 0; j < (MODE == 'C' ? 2 * NM : NM); ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < (MODE == 'C' ? 2 * NV : NV); ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 16]) : "v"(m), "v"(ad));
            }
        }
    }
    (void)nm; (void)nv;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + c[0][i] + c[1][i] + c[2][i] + c[3][i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NM, int NV> static void run(const char* name, float* out, int cus) {
    const int reps = 20000;
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(cus), dim3(1024), 0, 0, 500, out);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(cus), dim3(1024), 0, 0, reps, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    /* per SIMD and loop iteration of its 4 waves: 4 NM MFMAs and 4 NV VALU wave-instructions */
    printf("  %-44s %8.3f ms  %7.1f ns per iteration = %6.1f cycles@2.4 per SIMD per (NM MFMA + NV VALU of one wave)\n", name, ms, ms * 1e6 / reps, ms * 1e-3 * 2.4e9 / reps / 4);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    float* out; CHK(hipMalloc(&out, 16));
    int cus = p.multiProcessorCount;
    printf("4 waves per SIMD, NM = 4 MFMA 32x32x16 f16 + NV = 64 v_fma_f32 per wave and iteration (the tile of the broad phase)\n");
    run<'M', 4, 64>("M  MFMA only", out, cus);
    run<'V', 4, 64>("V  VALU only", out, cus);
    run<'A', 4, 64>("A  4 MFMA then 64 VALU, every wave", out, cus);
    run<'B', 4, 64>("B  interleaved 1 MFMA : 16 VALU, every wave", out, cus);
    run<'C', 4, 64>("C  even waves MFMA only, odd waves VALU only", out, cus);
    printf("NM = 4, NV = 32\n");
    run<'V', 4, 32>("V  VALU only", out, cus);
    run<'A', 4, 32>("A  4 MFMA then 32 VALU", out, cus);
    run<'B', 4, 32>("B  interleaved 1 : 8", out, cus);
    run<'C', 4, 32>("C  specialised waves", out, cus);
    printf("NM = 4, NV = 16\n");
    run<'V', 4, 16>("V  VALU only", out, cus);
    run<'A', 4, 16>("A  4 MFMA then 16 VALU", out, cus);
    run<'B', 4, 16>("B  interleaved 1 : 4", out, cus);
    run<'C', 4, 16>("C  specialised waves", out, cus);
    return 0;
}
