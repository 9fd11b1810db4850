// Do independent VALU instructions issue while an MFMA executes on gfx950?  Loop body: one v_mfma_f32_32x32x16_f16
// (8 passes) + NV independent v_fma_f32; time per iteration against NV, for 1, 2 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, int NM>
__global__ __launch_bounds__(256) void k(int reps, float* out, unsigned long long* clk) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(1.f - i * 0.1f); }
    f16v c0 = {}, c1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    const float m = 1.0001f + threadIdx.x * 1e-9f, ad = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (NM >= 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        if (NM >= 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 16]) : "v"(m), "v"(ad));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + c0[i] + c1[i];
    if (s == 12345.678f) out[0] = s;
}

template <int NV, int NM> static void run(float* out, unsigned long long* clk, int cus) {
    const int reps = 20000;
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
        hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(256), 0, 0, 200, out, clk);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(256), 0, 0, reps, out, clk);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h; CHK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        double ghz = (double)h / (ms * 1e-3) / 1e9 ;  // s_memtime ticks at 100 MHz? printed raw below
        (void)ghz;
        printf("  MFMA x%d + %2d VALU, %d waves/SIMD: %7.3f ms  -> %6.1f ns per iteration per wave-slot, %6.1f cycles@2.4 per SIMD per iteration\n", NM, NV, wps, ms,
               ms * 1e6 / reps, ms * 1e-3 * 2.4e9 / reps / wps);
    }
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    float* out; unsigned long long* clk;
    CHK(hipMalloc(&out, 16)); CHK(hipMalloc(&clk, 16));
    int cus = p.multiProcessorCount;
    run<0, 1>(out, clk, cus);
    run<8, 0>(out, clk, cus);
    run<16, 0>(out, clk, cus);
    run<4, 1>(out, clk, cus);
    run<8, 1>(out, clk, cus);
    run<12, 1>(out, clk, cus);
    run<16, 1>(out, clk, cus);
    run<32, 1>(out, clk, cus);
    run<32, 0>(out, clk, cus);
    run<32, 2>(out, clk, cus);
    return 0;
}
