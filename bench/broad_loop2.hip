// Microbenchmark, second round: the f32 conservative broad-phase loop with R rays per lane (register tiling, as in an
// n-body tile loop).  Each 16-byte record read from LDS (wave-uniform ds_read_b128) is used for R (ray, sphere) pairs.
// Shape under test = the product loop: 8 VALU per pair + one v_alignbit per pair + one unconditional ds_write per
// group of 16 per ray.  Reports nominal (2.4 GHz) cycles per wave-pair per SIMD; ideal is 18 (9 VALU x 2 cycles).
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o broad_loop2 broad_loop2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct RayC { float Dnx, Dny, Dnz, k1, O2x, O2y, O2z, nk2; };
#define H(r, s) __builtin_fmaf(r.Dnz, s.z, __builtin_fmaf(r.Dny, s.y, __builtin_fmaf(r.Dnx, s.x, -r.k1)))
#define QW(r, s) __builtin_fmaf(H(r, s), H(r, s), __builtin_fmaf(r.O2z, s.z, __builtin_fmaf(r.O2y, s.y, __builtin_fmaf(r.O2x, s.x, r.nk2 - s.w))))

template <int R, int MINW, int READS>
__global__ __launch_bounds__(256, MINW) void k_broad(const float4* S, int n, int reps, float* out) {
    __shared__ float4 tab[512];
    __shared__ uint32_t ent[4][R][16][64];
    for (int i = threadIdx.x; i < n; i += 256) tab[i] = S[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float t = threadIdx.x * 1e-3f;
    RayC ray[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ray[r] = {0.6f + t + r, 0.1f + t, 0.8f - t, 1e6f + t, t, 2.f + r + t, 3.f + t, -1e9f - t};
    uint32_t cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cnt[r] = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int g = 0; g < n / 16; ++g) {
            const float4* p = tab + g * 16;
            uint32_t acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = g;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 s[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) s[k] = READS ? p[half * 8 + k] : make_float4(t + k, t * 2, t - k, 1e3f);
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_alignbit(acc[r], __float_as_uint(QW(ray[r], s[k])), 31);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ent[w][r][cnt[r] & 15][lane] = acc[r];
                cnt[r] += ((acc[r] & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ray[r].nk2 -= 1.0f;
    }
    uint32_t tot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tot += cnt[r];
    if (tot == 123456) out[0] = (float)tot;
}

// Same loop, R = 1, but the eight chains of a half-group advance in lock step (stage by stage), so that two dependent
// VALU instructions are eight instructions apart instead of two.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_staged(const float4* S, int n, int reps, float* out) {
    __shared__ float4 tab[512];
    __shared__ uint32_t ent[4][16][64];
    for (int i = threadIdx.x; i < n; i += 256) tab[i] = S[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float t = threadIdx.x * 1e-3f;
    RayC r = {0.6f + t, 0.1f + t, 0.8f - t, 1e6f + t, t, 2.f + t, 3.f + t, -1e9f - t};
    uint32_t cnt = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int g = 0; g < n / 16; ++g) {
            const float4* p = tab + g * 16;
            uint32_t acc0 = g, acc1 = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 s[8];
                float h[8], e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) s[k] = p[half * 8 + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) { e[k] = r.nk2 - s[k].w; h[k] = __builtin_fmaf(r.Dnx, s[k].x, -r.k1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) { e[k] = __builtin_fmaf(r.O2x, s[k].x, e[k]); h[k] = __builtin_fmaf(r.Dny, s[k].y, h[k]); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) { e[k] = __builtin_fmaf(r.O2y, s[k].y, e[k]); h[k] = __builtin_fmaf(r.Dnz, s[k].z, h[k]); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = __builtin_fmaf(r.O2z, s[k].z, e[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = __builtin_fmaf(h[k], h[k], e[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    acc0 = __builtin_amdgcn_alignbit(acc0, __float_as_uint(e[k]), 31);
                    acc1 = __builtin_amdgcn_alignbit(acc1, __float_as_uint(e[k + 1]), 31);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t acc = (acc0 << 8) ^ acc1;
            ent[w][cnt & 15][lane] = acc;
            cnt += ((acc & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
        }
        r.nk2 -= 1.0f;
    }
    if (cnt == 123456) out[0] = (float)cnt;
}
template <int MINW> static void run_staged(const float4* dS, int n, float* out, int cus, int blocks_per_cu) {
    int blocks = cus * blocks_per_cu, reps = 1000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_staged<MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, 20, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_staged<MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, reps, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double pairs_per_simd = (double)blocks_per_cu * reps * n;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("staged    rays/lane 1, %d waves/SIMD: %8.3f ms, %6.2f nominal cycles per wave-pair per SIMD\n", blocks_per_cu, ms, cyc / pairs_per_simd);
}

template <int R, int MINW, int READS> static void run(const float4* dS, int n, float* out, int cus, int blocks_per_cu) {
    int blocks = cus * blocks_per_cu, reps = 1000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_broad<R, MINW, READS>), dim3(blocks), dim3(256), 0, 0, dS, n, 20, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_broad<R, MINW, READS>), dim3(blocks), dim3(256), 0, 0, dS, n, reps, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double pairs_per_simd = (double)blocks_per_cu * reps * n * R;  // wave-level (ray,sphere) evaluations per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    double tests = (double)blocks * 256 * reps * n * R;
    printf("%s rays/lane %d, %d waves/SIMD: %8.3f ms, %6.2f nominal cycles per wave-pair per SIMD (ideal 18), %7.1f G pair-tests/s\n",
           READS ? "LDS reads" : "no reads ", R, blocks_per_cu, ms, cyc / pairs_per_simd, tests / (ms * 1e-3) / 1e9);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int n = 496;
    std::vector<float4> S(n + 16);
    for (int i = 0; i < (int)S.size(); ++i) S[i] = make_float4(i * 0.01f, 0.2f, -i * 0.02f, 1.0f + i);
    float4* dS; float* out;
    CHK(hipMalloc(&dS, S.size() * sizeof(float4))); CHK(hipMalloc(&out, 16));
    CHK(hipMemcpy(dS, S.data(), S.size() * sizeof(float4), hipMemcpyHostToDevice));
    int cus = p.multiProcessorCount;
    run<1, 4, 1>(dS, n, out, cus, 4);
    run_staged<4>(dS, n, out, cus, 4);
    run_staged<4>(dS, n, out, cus, 2);
    run_staged<8>(dS, n, out, cus, 8);
    run<1, 4, 1>(dS, n, out, cus, 2);
    run<1, 4, 0>(dS, n, out, cus, 4);
    run<1, 8, 1>(dS, n, out, cus, 8);
    run<2, 4, 1>(dS, n, out, cus, 4);
    run<2, 2, 1>(dS, n, out, cus, 2);
    run<2, 2, 1>(dS, n, out, cus, 3);
    run<2, 2, 0>(dS, n, out, cus, 2);
    run<4, 2, 1>(dS, n, out, cus, 2);
    run<4, 1, 1>(dS, n, out, cus, 1);
    run<4, 2, 0>(dS, n, out, cus, 2);
    return 0;
}
