// What does a 32 x 32 -> 64 bit product cost on gfx950?  Philox4x32-10 needs twenty per block; hipcc compiles
// (uint64_t)a * b to v_mul_lo_u32 + v_mul_hi_u32 (two quarter-rate instructions).  v_mad_u64_u32 delivers both halves in
// ONE instruction.  Modes: 0 = v_fma_f32 chains (the yardstick), 1 = mul_lo + mul_hi pairs, 2 = v_mad_u64_u32 (inline asm),
// 3 = whole Philox blocks as the library writes them, 4 = whole Philox blocks with v_mad_u64_u32.
// Also checks the wave-wide DPP prefix sum used by the balanced narrow phase (row_shr + row_bcast) against a serial sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline uint64_t mad64(uint32_t a, uint32_t b) {
    uint64_t r;
    uint64_t carry; // VOP3B: the carry-out goes to an SGPR pair
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
    return r;
}
template <bool MAD> __device__ inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        uint32_t m0 = 0xD2511F53u, m1 = 0xCD9E8D57u;
        uint64_t p0, p1;
        if (MAD) { asm volatile("" : "+v"(m0), "+v"(m1)); p0 = mad64(m0, c0); p1 = mad64(m1, c2); }
        else { p0 = (uint64_t)m0 * c0; p1 = (uint64_t)m1 * c2; }
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t x[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i + seed; f[i] = (float)(threadIdx.x + i); }
    uint32_t m = 0xD2511F53u + seed;
    float fa = 1.0001f, fb = 0.5f;
    asm volatile("" : "+v"(m), "+v"(fa), "+v"(fb));
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], fa, fb);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { uint64_t p = (uint64_t)x[i] * m; x[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { uint64_t p = mad64(x[i], m); x[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }
        } else {
            uint32_t o[4];
            philox<MODE == 4>(x[0], x[1], x[2], x[3], seed, seed + 1, o);
            x[0] = o[0]; x[1] = o[1]; x[2] = o[2]; x[3] = o[3];
        }
    }
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += x[i] + (uint32_t)f[i];
    if (MODE >= 3) out[blockIdx.x * 256 + threadIdx.x] = t; // the two Philox variants must agree
    else if (t == 123u) out[0] = t;
}
template <int MODE> static double run(const char* name, int cus, int wps, uint32_t* out, double per_iter) {
    int iters = MODE >= 3 ? 2000 : 20000, blocks = cus * wps;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 7u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * per_iter);
    printf("%-34s %d waves/SIMD: %7.2f nominal cycles per %s per SIMD\n", name, wps, cyc, MODE >= 3 ? "Philox block" : MODE == 0 ? "v_fma_f32" : "64-bit product (+ xor)");
    return cyc;
}

// the scan of mirt_kernels.h::wave_scan_incl, copied here to be checked on the hardware
__device__ inline uint32_t dpp_add(uint32_t x, int ctrl, int row_mask) {
    return x; // placeholder, specialised below (the DPP control must be a compile-time constant)
}
__device__ inline uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1, 3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2, 3 */
    return x;
}
__global__ void scan_kernel(const uint32_t* in, uint32_t* out) { out[threadIdx.x] = wave_scan_incl(in[threadIdx.x]); }

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    uint32_t *out, *out2; CHK(hipMalloc(&out, 8 * 256 * 1024 * 4)); CHK(hipMalloc(&out2, 8 * 256 * 1024 * 4));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", p.multiProcessorCount, w, out, 8);
        run<1>("v_mul_lo_u32 + v_mul_hi_u32", p.multiProcessorCount, w, out, 8);
        run<2>("v_mad_u64_u32", p.multiProcessorCount, w, out, 8);
        run<3>("Philox4x32-10, mul_lo + mul_hi", p.multiProcessorCount, w, out, 1);
        run<4>("Philox4x32-10, v_mad_u64_u32", p.multiProcessorCount, w, out2, 1);
    }
    {   // the two Philox variants compute the same integers
        int n = p.multiProcessorCount * 4 * 256;
        uint32_t* a = (uint32_t*)malloc(n * 4); uint32_t* b = (uint32_t*)malloc(n * 4);
        CHK(hipMemcpy(a, out, n * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b, out2, n * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < n; ++i) bad += a[i] != b[i];
        printf("Philox variants agree on %d lanes: %s\n", n, bad ? "NO" : "yes");
    }
    {   // DPP scan against a serial prefix sum
        uint32_t h[64], r[64]; uint32_t *din, *dout; CHK(hipMalloc(&din, 256)); CHK(hipMalloc(&dout, 256));
        int bad = 0;
        for (int trial = 0; trial < 50; ++trial) {
            for (int i = 0; i < 64; ++i) h[i] = (uint32_t)(rand() % 9) * (trial % 3 == 0 ? 1u : (uint32_t)(rand() % 2));
            CHK(hipMemcpy(din, h, 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, 0, din, dout);
            CHK(hipMemcpy(r, dout, 256, hipMemcpyDeviceToHost));
            uint32_t s = 0; for (int i = 0; i < 64; ++i) { s += h[i]; bad += r[i] != s; }
        }
        printf("DPP wave scan (row_shr 1,2,4,8 + row_bcast 15, 31): %s\n", bad ? "WRONG" : "ok");
    }
    return 0;
}
