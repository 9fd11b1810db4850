// FP32 / FP64 vector-ALU peak microbenchmark for gfx950 ("FP32_peak measured on the box").
// Independent FMA chains held in registers; reports TFLOP/s for v_fma_f32, v_pk_fma_f32 and v_fma_f64 at
// 1, 2, 4 and 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_peak valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float float2v __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ITERS> __global__ void k_fma32(float* out, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 123.456f) out[0] = s;
}
template <int ITERS> __global__ void k_pkfma32(float* out, float a, float b) {
    float2v x[16];
    float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = float2v{(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    if (s == 123.456f) out[0] = s;
}
template <int ITERS> __global__ void k_fma64(float* out, double a, double b) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (double)(threadIdx.x + i);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 123.456) out[0] = (float)s;
}
// mixed: the instruction mix of the sphere test (3 sub, 1 mul, 5 fma, 1 mul, 1 fma) on 4 independent spheres
template <int ITERS> __global__ void k_mix(float* out, float a, float b) {
    float ox = threadIdx.x, oy = a, oz = b, dx = 0.3f, dy = 0.4f, dz = a * b, acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float cx = a + i + it, cy = b - i, cz = a * i;
            float x = cx - ox, y = cy - oy, z = cz - oz;
            float h = __builtin_fmaf(dz, z, __builtin_fmaf(dy, y, dx * x));
            float c = __builtin_fmaf(x, x, __builtin_fmaf(y, y, __builtin_fmaf(z, z, -b)));
            float d = __builtin_fmaf(h, h, -(a * c));
            acc = fmaxf(acc, d);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <class F> static double run(F launch, int blocks, int threads, double flop_per_thread) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(blocks, threads);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch(blocks, threads);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    return 5.0 * flop_per_thread * blocks * threads / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz LDS/block %zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    float* out; CHK(hipMalloc(&out, 4));
    const int IT = 4096;
    int cus = p.multiProcessorCount;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps, threads = 256; // 4 waves per block -> one per SIMD; wps blocks per CU
        double f32 = run([&](int b, int t) { hipLaunchKernelGGL(k_fma32<IT>, dim3(b), dim3(t), 0, 0, out, 1.0001f, 0.5f); }, blocks, threads, 2.0 * 16 * IT);
        double pk = run([&](int b, int t) { hipLaunchKernelGGL(k_pkfma32<IT>, dim3(b), dim3(t), 0, 0, out, 1.0001f, 0.5f); }, blocks, threads, 4.0 * 16 * IT);
        double f64 = run([&](int b, int t) { hipLaunchKernelGGL(k_fma64<IT>, dim3(b), dim3(t), 0, 0, out, 1.0001, 0.5); }, blocks, threads, 2.0 * 16 * IT);
        double mix = run([&](int b, int t) { hipLaunchKernelGGL(k_mix<IT>, dim3(b), dim3(t), 0, 0, out, 1.0001f, 0.5f); }, blocks, threads, 18.0 * 4 * IT);
        printf("waves/SIMD %d : v_fma_f32 %.1f TF | v_pk_fma_f32 %.1f TF | v_fma_f64 %.1f TF | sphere-mix (18 flop/test) %.1f TF\n", wps, f32, pk, f64, mix);
    }
    return 0;
}
