// Probe for a broad phase on the matrix pipe (v_mfma_f32_32x32x16_f16 with split-f16 operands).
//   part 1: operand / result layout, f16 subnormal inputs, accumulation error against a double reference
//   part 2: throughput of the candidate loop: per tile of 32 spheres two ds_read_b128 (sphere operands), four MFMAs
//           (G and E for the two halves of the wave's 64 rays), 32 x (fma + alignbit) per lane, one permlane32 swap,
//           two entry-list writes
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(const h8* a, const h8* b, float* d) {
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = c[r];
}

template <int THREADS, int EU, int MODE>
__global__ __launch_bounds__(THREADS, EU) void loop_kernel(const h8* GS, const h8* ES, int ntiles, int reps, float* out, unsigned long long* clk) {
    __shared__ h8 gs[16 * 64];
    __shared__ h8 es[16 * 64];
    __shared__ uint32_t ent[THREADS / 64][16][64];
    for (int i = threadIdx.x; i < ntiles * 64; i += THREADS) { gs[i] = GS[i]; es[i] = ES[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    h8 bgx, bgy, bex, bey;
    for (int k_ = 0; k_ < 8; ++k_) {
        bgx[k_] = (_Float16)(0.01f * (lane + k_));
        bgy[k_] = (_Float16)(0.02f * (lane - k_));
        bex[k_] = (_Float16)(0.5f + 0.01f * k_);
        bey[k_] = (_Float16)(-0.25f * lane + k_);
    }
    uint32_t cnt = 0;
    f16v fg, fe;
    for (int r = 0; r < 16; ++r) { fg[r] = 0.1f * r + lane; fe[r] = -0.3f * r - lane; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < ntiles; ++t) {
            const h8 ag = gs[t * 64 + lane], ae = es[t * 64 + lane];
            const f16v z = {};
            uint32_t mx = 2 * t, my = 2 * t + 1;
            {
                f16v g, e;
                if (MODE != 2) {
                    g = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, bgx, z, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_32x32x16_f16(ae, bex, z, 0, 0, 0);
                } else {
                    g = fg; e = fe;
                    asm volatile("" : "+v"(g), "+v"(e) : "v"(ag), "v"(ae));
                }
                if (MODE == 1) asm volatile("" ::"v"(g), "v"(e));
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = __builtin_amdgcn_alignbit(mx, __float_as_uint(__builtin_fmaf(g[r], g[r], e[r])), 31);
                }
            }
            {
                f16v g, e;
                if (MODE != 2) {
                    g = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, bgy, z, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_32x32x16_f16(ae, bey, z, 0, 0, 0);
                } else {
                    g = fg; e = fe;
                    asm volatile("" : "+v"(g), "+v"(e) : "v"(ag), "v"(ae));
                }
                if (MODE == 1) asm volatile("" ::"v"(g), "v"(e));
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) my = __builtin_amdgcn_alignbit(my, __float_as_uint(__builtin_fmaf(g[r], g[r], e[r])), 31);
                }
            }
            auto sw = __builtin_amdgcn_permlane32_swap(mx, my, false, false);
            mx = sw[0]; my = sw[1];
            ent[w][cnt & 15][lane] = mx;
            cnt += ((mx & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
            ent[w][cnt & 15][lane] = my;
            cnt += ((my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
        }
        bex[0] += (_Float16)0.001f;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (cnt == 123456789) out[0] = (float)cnt;
}

template <int THREADS, int EU, int MODE> static void run(const h8* dG, const h8* dE, int ntiles, float* out, unsigned long long* clk, int cus, int waves_per_simd) {
    int blocks = cus * waves_per_simd * 4 / (THREADS / 64), reps = 2000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop_kernel<THREADS, EU, MODE>), dim3(blocks), dim3(THREADS), 0, 0, dG, dE, ntiles, 20, out, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop_kernel<THREADS, EU, MODE>), dim3(blocks), dim3(THREADS), 0, 0, dG, dE, ntiles, reps, out, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double pairs_per_simd = (double)waves_per_simd * reps * ntiles * 32;
    printf("matrix-pipe broad loop mode %d %4d thr/block, %d waves/SIMD: %8.3f ms  %6.2f cycles@2.4 per (wave, sphere) per SIMD  (%.3f GHz)\n", MODE, THREADS,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / pairs_per_simd, ghz);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    // ---- part 1 ----
    {
        std::vector<_Float16> A(32 * 16), B(16 * 32);  // A[i][k], B[k][j]
        srand(7);
        auto rnd = [] { return (float)(rand() / (double)RAND_MAX) * 2.f - 1.f; };
        for (int i = 0; i < 32; ++i)
            for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)(rnd() * (k % 3 == 0 ? 100.f : k % 3 == 1 ? 0.05f : 1.f));
        for (int k = 0; k < 16; ++k)
            for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)(rnd() * (k % 2 ? 30.f : 0.3f));
        // row 0 / col 0: subnormal probes.  A[0][*] = 0 except A[0][0] = 2^-20 (subnormal); B[0][0] = 1024 -> 2^-10 if kept
        for (int k = 0; k < 16; ++k) A[0 * 16 + k] = (_Float16)0.f;
        A[0] = (_Float16)9.5367431640625e-07f;  // 2^-20
        for (int k = 0; k < 16; ++k) B[k * 32 + 0] = (_Float16)0.f;
        B[0] = (_Float16)1024.f;
        // row 1 / col 1: cancellation: 1000*1 + (-1000)*1 + 2^-10 * 1
        for (int k = 0; k < 16; ++k) { A[1 * 16 + k] = (_Float16)0.f; B[k * 32 + 1] = (_Float16)0.f; }
        A[16 + 0] = (_Float16)1000.f; A[16 + 1] = (_Float16)-1000.f; A[16 + 2] = (_Float16)0.0009765625f;
        B[0 * 32 + 1] = (_Float16)1.f; B[1 * 32 + 1] = (_Float16)1.f; B[2 * 32 + 1] = (_Float16)1.f;
        std::vector<h8> a(64), b(64);
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 8; ++k) {
                a[l][k] = A[(l % 32) * 16 + 8 * (l / 32) + k];
                b[l][k] = B[(8 * (l / 32) + k) * 32 + l % 32];
            }
        h8 *da, *db; float* dd;
        CHK(hipMalloc(&da, 64 * sizeof(h8))); CHK(hipMalloc(&db, 64 * sizeof(h8))); CHK(hipMalloc(&dd, 64 * 16 * 4));
        CHK(hipMemcpy(da, a.data(), 64 * sizeof(h8), hipMemcpyHostToDevice));
        CHK(hipMemcpy(db, b.data(), 64 * sizeof(h8), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<float> d(64 * 16);
        CHK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst_rel_sumabs = 0; int bad_layout = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;
                double ref = 0, sumabs = 0;
                for (int k = 0; k < 16; ++k) { double t = (double)A[i * 16 + k] * (double)B[k * 32 + j]; ref += t; sumabs += fabs(t); }
                double err = fabs((double)d[l * 16 + r] - ref);
                if (sumabs > 0 && err / sumabs > 1e-3) ++bad_layout;
                if (err > worst) worst = err;
                if (sumabs > 0 && err / sumabs > worst_rel_sumabs) worst_rel_sumabs = err / sumabs;
            }
        printf("layout D[8(r/4)+4(l/32)+r%%4][l%%32]: %d mismatches; max |err| %.3g; max err / sum|terms| = %.3g = 2^%.2f\n", bad_layout, worst,
               worst_rel_sumabs, log2(worst_rel_sumabs > 0 ? worst_rel_sumabs : 1e-300));
        printf("subnormal input 2^-20 * 1024 -> %.6g (kept = %.6g)\n", d[0 * 16 + 0], 0.0009765625);
        printf("cancellation 1000 - 1000 + 2^-10 -> %.6g (exact = %.6g)\n", d[1 * 16 + 1], 0.0009765625);
    }
    // ---- part 2 ----
    int ntiles = 16;
    std::vector<h8> G(ntiles * 64), E(ntiles * 64);
    for (size_t i = 0; i < G.size(); ++i)
        for (int k = 0; k < 8; ++k) { G[i][k] = (_Float16)(0.001f * (i % 97) + k); E[i][k] = (_Float16)(1.f - 0.002f * (i % 89) * k); }
    h8 *dG, *dE; float* out; unsigned long long* clk;
    CHK(hipMalloc(&dG, G.size() * sizeof(h8))); CHK(hipMalloc(&dE, E.size() * sizeof(h8))); CHK(hipMalloc(&out, 16)); CHK(hipMalloc(&clk, 16));
    CHK(hipMemcpy(dG, G.data(), G.size() * sizeof(h8), hipMemcpyHostToDevice));
    CHK(hipMemcpy(dE, E.data(), E.size() * sizeof(h8), hipMemcpyHostToDevice));
    int cus = p.multiProcessorCount;
    printf("mode 0 = MFMA + VALU, 1 = MFMA only, 2 = VALU only\n");
    run<1024, 4, 0>(dG, dE, ntiles, out, clk, cus, 4);
    run<1024, 4, 1>(dG, dE, ntiles, out, clk, cus, 4);
    run<1024, 4, 2>(dG, dE, ntiles, out, clk, cus, 4);
    run<512, 2, 0>(dG, dE, ntiles, out, clk, cus, 2);
    run<512, 2, 1>(dG, dE, ntiles, out, clk, cus, 2);
    run<512, 2, 2>(dG, dE, ntiles, out, clk, cus, 2);
    run<256, 1, 0>(dG, dE, ntiles, out, clk, cus, 1);
    run<256, 1, 1>(dG, dE, ntiles, out, clk, cus, 1);
    run<256, 1, 2>(dG, dE, ntiles, out, clk, cus, 1);
    return 0;
}
