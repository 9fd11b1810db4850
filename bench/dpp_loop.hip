// Microbenchmark, fourth round: broad-phase loop with the sphere records delivered by DPP row broadcasts.
// ONE ds_read_b128 per lane fetches the 16 records of a group (lane L of every 16-lane row holds record L mod 16);
// each (ray, sphere) test then reads "record j" through the DPP control row_newbcast:j on the VOP2 operand, so the
// broadcast costs neither LDS bandwidth nor a VALU slot.  q = (Dn.M)^2 + 2P.M + (c_ray - w), P = O - (Dn.O) Dn:
//   v_mul_f32_dpp / 2 x v_fmac_f32_dpp  (Dn.M)      3
//   v_subrev_f32_dpp                    c_ray - w   1
//   3 x v_fmac_f32_dpp                  + 2P.M      3
//   v_fma_f32                           a*a + e     1     = 8 VALU per pair, + v_alignbit for the sign
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o dpp_loop dpp_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define STR2(x) #x
#define STR(x) STR2(x)
// one (ray, record J) test; X,Y,Z,W = the four record VGPRs of the group
#define PAIR(J)                                                                                                     \
    {                                                                                                               \
        float a_, e_;                                                                                               \
        asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "=v"(a_) : "v"(X), "v"(Dnx));          \
        asm volatile("v_subrev_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "=v"(e_) : "v"(W), "v"(cray));     \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "+v"(a_) : "v"(Y), "v"(Dny));        \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "+v"(e_) : "v"(X), "v"(P2x));        \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "+v"(a_) : "v"(Z), "v"(Dnz));        \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "+v"(e_) : "v"(Y), "v"(P2y));        \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" STR(J) " row_mask:0xf bank_mask:0xf" : "+v"(e_) : "v"(Z), "v"(P2z));        \
        float q_ = __builtin_fmaf(a_, a_, e_);                                                                      \
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(q_), 31);                                              \
    }

template <int THREADS, bool CHECK>
__global__ __launch_bounds__(THREADS) void k(const float4* S, int n, int reps, uint32_t* out, unsigned long long* clk) {
    __shared__ float4 tab[512];
    __shared__ uint32_t ent[THREADS / 64][16][64];
    for (int i = threadIdx.x; i < n; i += THREADS) tab[i] = S[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float t = CHECK ? lane * 0.37f : threadIdx.x * 1e-3f;
    float Dnx = 0.6f, Dny = 0.48f, Dnz = 0.64f;                    // unit vector
    float Ox = 1.0f + t, Oy = 2.0f - 0.1f * t, Oz = -3.0f + 0.2f * t;
    float k1 = Dnx * Ox + Dny * Oy + Dnz * Oz;
    float P2x = 2.f * (Ox - k1 * Dnx), P2y = 2.f * (Oy - k1 * Dny), P2z = 2.f * (Oz - k1 * Dnz);
    float cray = k1 * k1 - (Ox * Ox + Oy * Oy + Oz * Oz);
    uint32_t cnt = 0, sum = 0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int g = 0; g < n / 16; ++g) {
            const float4 rec = tab[g * 16 + (lane & 15)];
            const float X = rec.x, Y = rec.y, Z = rec.z, W = rec.w;
            uint32_t acc = g;
            PAIR(0) PAIR(1) PAIR(2) PAIR(3) PAIR(4) PAIR(5) PAIR(6) PAIR(7)
            PAIR(8) PAIR(9) PAIR(10) PAIR(11) PAIR(12) PAIR(13) PAIR(14) PAIR(15)
            ent[w][cnt & 15][lane] = acc;
            cnt += ((acc & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
            if (CHECK) out[(g * THREADS + threadIdx.x)] = acc;
            sum += acc;
        }
        cray -= 1e-3f;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (!CHECK && cnt + sum == 123456) out[0] = cnt;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int n = 496;
    std::vector<float4> S(n + 16);
    for (int i = 0; i < (int)S.size(); ++i) {
        float x = (i % 22) - 11.f, y = 0.2f, z = (i / 22) - 11.f, rho = 0.45f;
        S[i] = make_float4(x, y, z, x * x + y * y + z * z - rho * rho);
    }
    float4* dS; uint32_t* out; unsigned long long* clk;
    CHK(hipMalloc(&dS, S.size() * sizeof(float4))); CHK(hipMalloc(&out, 31 * 64 * 4 + 64)); CHK(hipMalloc(&clk, 16));
    CHK(hipMemcpy(dS, S.data(), S.size() * sizeof(float4), hipMemcpyHostToDevice));
    // ---- correctness of the DPP formulation against a host evaluation (one wave)
    hipLaunchKernelGGL((k<64, true>), dim3(1), dim3(64), 0, 0, dS, n, 1, out, clk);
    CHK(hipDeviceSynchronize());
    std::vector<uint32_t> h(31 * 64);
    CHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, cand = 0;
    for (int g = 0; g < 31; ++g)
        for (int lane = 0; lane < 64; ++lane) {
            float t = lane * 0.37f;
            double Dn[3] = {0.6f, 0.48f, 0.64f}, O[3] = {1.0f + t, 2.0f - 0.1f * t, -3.0f + 0.2f * t};
            for (int j = 0; j < 16; ++j) {
                const float4& s = S[g * 16 + j];
                double M[3] = {s.x, s.y, s.z};
                double hh = 0, d2 = 0;
                for (int c = 0; c < 3; ++c) { hh += Dn[c] * (M[c] - O[c]); d2 += (M[c] - O[c]) * (M[c] - O[c]); }
                double q = hh * hh - d2 + 0.45 * 0.45;
                int bit = (h[g * 64 + lane] >> (15 - j)) & 1;  // 1 = sign set = not a candidate
                if (std::fabs(q) > 1e-2 * (1 + d2) * 1e-3) { if ((q < 0) != (bit == 1)) ++bad; }
                if (q >= 0) ++cand;
            }
            if ((h[g * 64 + lane] >> 16) != (uint32_t)g) ++bad;
        }
    printf("DPP formulation check: %d mismatches, %d candidates of %d pairs\n", bad, cand, 31 * 64 * 16);
    // ---- throughput
    int cus = p.multiProcessorCount;
    for (int wps : {4, 2, 1, 6, 8}) {
        int blocks = cus * wps, reps = 1000;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<256, false>), dim3(blocks), dim3(256), 0, 0, dS, n, 20, out, clk);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<256, false>), dim3(blocks), dim3(256), 0, 0, dS, n, reps, out, clk);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hc[2]; CHK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;
        printf("DPP broadcast loop, %d waves/SIMD: %8.3f ms  %6.2f ns-cycles@2.4 per wave-pair per SIMD  (%.3f GHz)\n", wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)wps * reps * n), ghz);
    }
    return 0;
}
