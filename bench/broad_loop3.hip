// Microbenchmark, third round: what bounds the f32 broad-phase loop at 4 waves/SIMD?
//   mode 0: the product loop (16 broadcast ds_read_b128 + 16 x (8 FMA-class + 1 alignbit) per iteration)
//   mode 1: the same VALU work on records that stay in registers (made opaque to the optimiser, no LDS traffic)
//   mode 2: the LDS reads alone (values consumed by an empty asm)
//   mode 3: mode 0 with s_setprio raised around the VALU burst
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o broad_loop3 broad_loop3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct RayC { float Dnx, Dny, Dnz, k1, O2x, O2y, O2z, nk2; };
#define H(r, s) __builtin_fmaf(r.Dnz, s.z, __builtin_fmaf(r.Dny, s.y, __builtin_fmaf(r.Dnx, s.x, -r.k1)))
#define QW(r, s) __builtin_fmaf(H(r, s), H(r, s), __builtin_fmaf(r.O2z, s.z, __builtin_fmaf(r.O2y, s.y, __builtin_fmaf(r.O2x, s.x, r.nk2 - s.w))))

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float4* S, int n, int reps, float* out, unsigned long long* clk) {
    __shared__ float4 tab[512];
    __shared__ uint32_t ent[THREADS / 64][16][64];
    for (int i = threadIdx.x; i < n; i += THREADS) tab[i] = S[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float t = threadIdx.x * 1e-3f;
    RayC r = {0.6f + t, 0.1f + t, 0.8f - t, 1e6f + t, t, 2.f + t, 3.f + t, -1e9f - t};
    uint32_t cnt = 0;
    float4 reg[8];
    for (int k_ = 0; k_ < 8; ++k_) reg[k_] = tab[k_ + lane % 3];
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int g = 0; g < n / 16; ++g) {
            const float4* p = tab + g * 16;
            uint32_t acc = g;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 s[8];
#pragma unroll
                for (int k_ = 0; k_ < 8; ++k_) {
                    if (MODE == 1) {
                        s[k_] = reg[k_];
                        asm volatile("" : "+v"(s[k_].x), "+v"(s[k_].y), "+v"(s[k_].z), "+v"(s[k_].w));
                    } else {
                        s[k_] = p[half * 8 + k_];
                    }
                }
                if (MODE == 3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int k_ = 0; k_ < 8; ++k_) {
                    if (MODE == 2) asm volatile("" ::"v"(s[k_].x), "v"(s[k_].y), "v"(s[k_].z), "v"(s[k_].w));
                    else acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(QW(r, s[k_])), 31);
                }
                if (MODE == 3) __builtin_amdgcn_s_setprio(0);
            }
            if (MODE != 2) {
                ent[w][cnt & 15][lane] = acc;
                cnt += ((acc & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
            }
        }
        r.nk2 -= 1.0f;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (cnt == 123456) out[0] = (float)cnt;
}

template <int MODE, int THREADS> static void run(const char* name, const float4* dS, int n, float* out, unsigned long long* clk, int cus, int waves_per_simd) {
    int blocks = cus * waves_per_simd * 4 / (THREADS / 64), reps = 1000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, dS, n, 20, out, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, dS, n, reps, out, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;  // s_memtime ticks per second of s_memrealtime
    double pairs_per_simd = (double)waves_per_simd * reps * n;
    double sec = ms * 1e-3;
    printf("%-28s %d thr/block, %d waves/SIMD: %8.3f ms  %6.2f ns-cycles@2.4 per wave-pair per SIMD  (s_memtime/s_memrealtime -> %.3f GHz)\n", name, THREADS,
           waves_per_simd, ms, sec * 2.4e9 / pairs_per_simd, ghz);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int n = 496;
    std::vector<float4> S(n + 16);
    for (int i = 0; i < (int)S.size(); ++i) S[i] = make_float4(i * 0.01f, 0.2f, -i * 0.02f, 1.0f + i);
    float4* dS; float* out; unsigned long long* clk;
    CHK(hipMalloc(&dS, S.size() * sizeof(float4))); CHK(hipMalloc(&out, 16)); CHK(hipMalloc(&clk, 16));
    CHK(hipMemcpy(dS, S.data(), S.size() * sizeof(float4), hipMemcpyHostToDevice));
    int cus = p.multiProcessorCount;
    for (int w : {4, 2, 1}) {
        run<0, 256>("LDS + VALU (product loop)", dS, n, out, clk, cus, w);
        run<1, 256>("VALU only (regs)", dS, n, out, clk, cus, w);
        run<2, 256>("LDS reads only", dS, n, out, clk, cus, w);
        run<3, 256>("LDS + VALU, setprio", dS, n, out, clk, cus, w);
    }
    run<0, 1024>("LDS + VALU (product loop)", dS, n, out, clk, cus, 4);
    run<1, 1024>("VALU only (regs)", dS, n, out, clk, cus, 4);
    return 0;
}
