// Microbenchmark of the f32 conservative broad-phase loop in isolation (no candidates are ever found, so the rare
// branch is never taken): cycles per (ray, sphere) pair per SIMD for several loop shapes and occupancies.
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -o broad_loop broad_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) CF4;

#define H(s) __builtin_fmaf(Dnz, s.z, __builtin_fmaf(Dny, s.y, __builtin_fmaf(Dnx, s.x, -k1)))
#define QW(s) __builtin_fmaf(H(s), H(s), __builtin_fmaf(O2z, s.z, __builtin_fmaf(O2y, s.y, __builtin_fmaf(O2x, s.x, nk2 - s.w))))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define FI(x) __float_as_int(x)

template <int GROUP, int MINW>
__global__ __launch_bounds__(256, MINW) void k_broad(const float4* S, int n, int reps, float* out) {
    CF4* Sc = (CF4*)S;
    float t = threadIdx.x * 1e-3f;
    float Dnx = 0.6f + t, Dny = 0.0f, Dnz = 0.8f - t, k1 = 1e6f, O2x = t, O2y = 2.f, O2z = 3.f, nk2 = -1e9f - t;
    int hits = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        CF4* p = Sc;
        for (int g = 0; g < n / GROUP; ++g, p += GROUP) {
            int m;
            if (GROUP == 8) {
                f4v s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3], s4 = p[4], s5 = p[5], s6 = p[6], s7 = p[7];
                float q0 = QW(s0), q1 = QW(s1), q2 = QW(s2), q3 = QW(s3), q4 = QW(s4), q5 = QW(s5), q6 = QW(s6), q7 = QW(s7);
                m = IMAX(IMAX(FI(q0), FI(q1)), FI(q2));
                m = IMAX(IMAX(m, FI(q3)), FI(q4));
                m = IMAX(IMAX(m, FI(q5)), FI(q6));
                m = IMAX(m, FI(q7));
            } else if (GROUP == 16) {
                f4v s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3], s4 = p[4], s5 = p[5], s6 = p[6], s7 = p[7];
                f4v s8 = p[8], s9 = p[9], sa = p[10], sb = p[11], sc = p[12], sd = p[13], se = p[14], sf = p[15];
                float q0 = QW(s0), q1 = QW(s1), q2 = QW(s2), q3 = QW(s3), q4 = QW(s4), q5 = QW(s5), q6 = QW(s6), q7 = QW(s7);
                float q8 = QW(s8), q9 = QW(s9), qa = QW(sa), qb = QW(sb), qc = QW(sc), qd = QW(sd), qe = QW(se), qf = QW(sf);
                m = IMAX(IMAX(FI(q0), FI(q1)), FI(q2));
                m = IMAX(IMAX(m, FI(q3)), FI(q4));
                m = IMAX(IMAX(m, FI(q5)), FI(q6));
                m = IMAX(IMAX(m, FI(q7)), FI(q8));
                m = IMAX(IMAX(m, FI(q9)), FI(qa));
                m = IMAX(IMAX(m, FI(qb)), FI(qc));
                m = IMAX(IMAX(m, FI(qd)), FI(qe));
                m = IMAX(m, FI(qf));
            } else {
                f4v s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
                float q0 = QW(s0), q1 = QW(s1), q2 = QW(s2), q3 = QW(s3);
                m = IMAX(IMAX(FI(q0), FI(q1)), IMAX(FI(q2), FI(q3)));
            }
            if (__ballot(m >= 0) != 0ull) { ++hits; nk2 -= 1.0f; }
        }
        nk2 -= 1.0f; // keep iterations distinct
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (float)(t1 - t0); out[2] = (float)(r1 - r0); }
    if (hits == 123456) out[0] = (float)hits;
}

// LDS variant: the records sit in a block-shared LDS table and are read as wave-uniform broadcasts into VGPRs
template <int GROUP, int MINW>
__global__ __launch_bounds__(256, MINW) void k_broad_lds(const float4* S, int n, int reps, float* out) {
    __shared__ float4 tab[512];
    for (int i = threadIdx.x; i < n; i += 256) tab[i] = S[i];
    __syncthreads();
    float t = threadIdx.x * 1e-3f;
    float Dnx = 0.6f + t, Dny = 0.0f, Dnz = 0.8f - t, k1 = 1e6f, O2x = t, O2y = 2.f, O2z = 3.f, nk2 = -1e9f - t;
    int hits = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        for (int g = 0; g < n / GROUP; ++g) {
            const float4* p = tab + g * GROUP;
            int m;
            if (GROUP == 8) {
                float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3], s4 = p[4], s5 = p[5], s6 = p[6], s7 = p[7];
                float q0 = QW(s0), q1 = QW(s1), q2 = QW(s2), q3 = QW(s3), q4 = QW(s4), q5 = QW(s5), q6 = QW(s6), q7 = QW(s7);
                m = IMAX(IMAX(FI(q0), FI(q1)), FI(q2));
                m = IMAX(IMAX(m, FI(q3)), FI(q4));
                m = IMAX(IMAX(m, FI(q5)), FI(q6));
                m = IMAX(m, FI(q7));
            } else {
                float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
                float q0 = QW(s0), q1 = QW(s1), q2 = QW(s2), q3 = QW(s3);
                m = IMAX(IMAX(FI(q0), FI(q1)), IMAX(FI(q2), FI(q3)));
            }
            if (__ballot(m >= 0) != 0ull) { ++hits; nk2 -= 1.0f; }
        }
        nk2 -= 1.0f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (float)(t1 - t0); out[2] = (float)(r1 - r0); }
    if (hits == 123456) out[0] = (float)hits;
}

template <int GROUP, int MINW, bool LDS> static void run(const float4* dS, int n, float* out, int cus, int waves_per_simd) {
    int blocks = cus * waves_per_simd, reps = 2000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    if (LDS) hipLaunchKernelGGL((k_broad_lds<GROUP, MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, 50, out);
    else hipLaunchKernelGGL((k_broad<GROUP, MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, 50, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    if (LDS) hipLaunchKernelGGL((k_broad_lds<GROUP, MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, reps, out);
    else hipLaunchKernelGGL((k_broad<GROUP, MINW>), dim3(blocks), dim3(256), 0, 0, dS, n, reps, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    float h[3]; CHK(hipMemcpy(h, out, 12, hipMemcpyDeviceToHost));
    double ghz = h[1] / (h[2] / 100e6) / 1e9; // s_memtime ticks per second of s_memrealtime (100 MHz)
    printf("[clock %.2f GHz] %s ", ghz, LDS ? "LDS " : "SMEM");
    double pairs_per_simd = (double)waves_per_simd * reps * n;              // wave-level (ray,sphere) evaluations per SIMD
    double cyc = ms * 1e-3 * 2.4e9;                                           // at the nominal 2.4 GHz
    double tests = (double)blocks * 256 * reps * n;
    printf("group %2d, %d waves/SIMD: %.3f ms, %.2f nominal cycles per wave-pair per SIMD (8 VALU -> ideal 16), %.1f G pair-tests/s, algorithmic(27 flop) %.1f TF\n",
           GROUP, waves_per_simd, ms, cyc / pairs_per_simd, tests / (ms * 1e-3) / 1e9, tests * 27 / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int n = 496; // multiple of 16
    std::vector<float4> S(n + 16);
    for (int i = 0; i < (int)S.size(); ++i) S[i] = make_float4(i * 0.01f, 0.2f, -i * 0.02f, 1.0f + i);
    float4* dS; float* out;
    CHK(hipMalloc(&dS, S.size() * sizeof(float4))); CHK(hipMalloc(&out, 16));
    CHK(hipMemcpy(dS, S.data(), S.size() * sizeof(float4), hipMemcpyHostToDevice));
    int cus = p.multiProcessorCount;
    for (int w : {2, 3, 4, 5, 8}) {
        run<8, 1, false>(dS, n, out, cus, w);
        run<4, 1, true>(dS, n, out, cus, w);
        run<8, 1, true>(dS, n, out, cus, w);
    }
    return 0;
}
