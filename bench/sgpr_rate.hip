// Does a VALU instruction that reads a *different* SGPR every time issue slower than one that re-reads the same SGPR
// or only VGPRs?  16 independent FMA chains per lane, three operand patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct S16 { float v[16]; };
template <int MODE> __global__ __launch_bounds__(256) void k(S16 s, float* out, int iters) {
    float x[16], vv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x + i; vv[i] = s.v[i] + threadIdx.x * 1e-9f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], s.v[0], 0.5f);       // same SGPR every time
            else if (MODE == 1) x[i] = __builtin_fmaf(x[i], s.v[i], 0.5f);  // 16 different SGPRs
            else x[i] = __builtin_fmaf(x[i], vv[i], 0.5f);                  // VGPR operands only
        }
    }
    float t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += x[i];
    if (t == 123.f) out[0] = t;
}
template <int MODE> static void run(const char* name, int cus, int wps) {
    S16 s; for (int i = 0; i < 16; ++i) s.v[i] = 1.0f + i * 1e-4f;
    float* out; CHK(hipMalloc(&out, 4));
    int iters = 20000, blocks = cus * wps;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, out, 100);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, out, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_simd = (double)wps * iters * 16;
    printf("%-22s %d waves/SIMD: %.2f nominal cycles per VALU instruction per SIMD, %.1f TF\n", name, wps, ms * 1e-3 * 2.4e9 / instr_per_simd,
           (double)blocks * 256 * iters * 16 * 2 / (ms * 1e-3) / 1e12);
    CHK(hipFree(out));
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    for (int w : {2, 4, 8}) {
        run<0>("same SGPR", p.multiProcessorCount, w);
        run<1>("16 different SGPRs", p.multiProcessorCount, w);
        run<2>("VGPR operands only", p.multiProcessorCount, w);
    }
}
