// What does one tile of the matrix-pipe broad phase (mirt_kernels.h::mx_tile: 4 x v_mfma_f32_32x32x16_f16 + 32 x v_alignbit_b32 +
// permlane32_swap + an 8-byte LDS write + a ballot) cost per SIMD with 4 waves resident, and how much of that is the ORDER of
// the instructions?  The matrix pipe needs 4 x 32 = 128 cycles per tile and wave, the vector ALU ~50 instructions.
// Variants of the loop body, all computing the same masks:
//   0  as hipcc schedules mx_tile today: chain X (2 MFMAs, 16 dependent alignbits), then chain Y into the SAME accumulators
//   1  both chains issued first (two accumulator sets), then the 32 alignbits, X and Y interleaved
//   2  software pipeline over tiles: alignbits of X(t), MFMAs X(t+1), alignbits of Y(t), MFMAs Y(t+1)  (two accumulator sets)
//   3  as 0 with each chain of 16 alignbits split in two chains of 8 (+ 1 v_lshl_or)
//   4  as 2 with the split chains of 3
// Prints wave-cycles per tile (s_memtime = shader clock) and, from the launch time, ns per tile and SIMD: the two together give the
// clock the chip held during the launch (REPS=10000 for launches long enough to reach the steady state).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define TILES 32

__device__ inline uint32_t collect16(uint32_t m, const f16v& q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) m = __builtin_amdgcn_alignbit(m, __float_as_uint(q[r]), 31);
    return m;
}
__device__ inline uint32_t collect8x2(uint32_t m, const f16v& q) {
    uint32_t lo = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(q[r]), 31);
        lo = __builtin_amdgcn_alignbit(lo, __float_as_uint(q[r + 8]), 31);
    }
    return (m << 8) | lo;
}
template <int V> __global__ __launch_bounds__(1024) void k(const uint4* __restrict__ SMg, uint32_t* out, unsigned long long* cyc, int reps, float seed) {
    __shared__ uint4 SMs[TILES * 128];
    __shared__ uint2 ent[16][8][64];
    for (int i = threadIdx.x; i < TILES * 128; i += 1024) SMs[i] = SMg[i];
    __syncthreads();
    const h8* SM = (const h8*)SMs;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, sub = lane >> 5;
    h8 bx0, bx1, by0, by1;
    for (int i = 0; i < 8; ++i) {
        bx0[i] = (_Float16)(seed * (float)((lane * 7 + i) % 13) - 0.3f);
        bx1[i] = (_Float16)(seed * (float)((lane * 5 + i) % 11) - 0.2f);
        by0[i] = (_Float16)(seed * (float)((lane * 3 + i) % 7) - 0.25f);
        by1[i] = (_Float16)(seed * (float)((lane * 11 + i) % 5) - 0.1f);
    }
    const f16v zero = {};
    uint32_t acc = 0, cnt = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        if (V == 2 || V == 4) {
            h8 a0 = SM[lane], a1 = SM[64 + lane];
            f16v qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bx0, zero, 0, 0, 0);
            f16v qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, by0, zero, 0, 0, 0);
            qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bx1, qx, 0, 0, 0);
            qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, by1, qy, 0, 0, 0);
            for (uint32_t t = 0; t < TILES; ++t) {
                const uint32_t tn = (t + 1) & (TILES - 1);
                a0 = SM[tn * 128 + lane]; a1 = SM[tn * 128 + 64 + lane];
                uint32_t mx = 2 * t + sub, my = mx;
                mx = V == 4 ? collect8x2(mx, qx) : collect16(mx, qx);
                __builtin_amdgcn_sched_barrier(0);
                qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bx0, zero, 0, 0, 0);
                qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bx1, qx, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                my = V == 4 ? collect8x2(my, qy) : collect16(my, qy);
                __builtin_amdgcn_sched_barrier(0);
                qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, by0, zero, 0, 0, 0);
                qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, by1, qy, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                auto sw = __builtin_amdgcn_permlane32_swap(mx, my, false, false);
                mx = sw[0]; my = sw[1];
                ent[w][cnt & 7u][lane] = make_uint2(mx, my);
                cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                if (__ballot(cnt > 1000000u) != 0ull) break;
                acc ^= mx + my;
            }
        } else {
            for (uint32_t t = 0; t < TILES; ++t) {
                const h8 a0 = SM[t * 128 + lane], a1 = SM[t * 128 + 64 + lane];
                uint32_t mx = 2 * t + sub, my = mx;
                if (V == 1) {
                    f16v qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bx0, zero, 0, 0, 0);
                    f16v qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, by0, zero, 0, 0, 0);
                    qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bx1, qx, 0, 0, 0);
                    qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, by1, qy, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        mx = __builtin_amdgcn_alignbit(mx, __float_as_uint(qx[r]), 31);
                        my = __builtin_amdgcn_alignbit(my, __float_as_uint(qy[r]), 31);
                    }
                } else {
                    f16v qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bx0, zero, 0, 0, 0);
                    qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bx1, qx, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mx = V == 3 ? collect8x2(mx, qx) : collect16(mx, qx);
                    __builtin_amdgcn_sched_barrier(0);
                    f16v qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, by0, zero, 0, 0, 0);
                    qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, by1, qy, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    my = V == 3 ? collect8x2(my, qy) : collect16(my, qy);
                }
                auto sw = __builtin_amdgcn_permlane32_swap(mx, my, false, false);
                mx = sw[0]; my = sw[1];
                ent[w][cnt & 7u][lane] = make_uint2(mx, my);
                cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                if (__ballot(cnt > 1000000u) != 0ull) break;
                acc ^= mx + my;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = acc + cnt;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}
template <int V> static void run(const uint4* SM, uint32_t* out, unsigned long long* cyc, std::vector<uint32_t>& ref) {
    const int blocks = getenv("BLOCKS") ? atoi(getenv("BLOCKS")) : 256, reps = getenv("REPS") ? atoi(getenv("REPS")) : 400;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(1024), 0, 0, SM, out, cyc, 4, 0.01f);
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(cyc, 0, 8));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(1024), 0, 0, SM, out, cyc, reps, 0.01f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> h(blocks * 1024);
    CHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    if (ref.empty()) ref = h;
    const double tiles = (double)reps * TILES;
    printf("variant %d: %8.1f wave-cycles per tile (4 waves per SIMD -> %6.1f SIMD-cycles), launch %7.3f ms = %6.1f ns per tile and SIMD -> shader clock %4.2f GHz; masks %s\n", V,
           (double)c / (blocks * 16.0) / tiles, (double)c / (blocks * 16.0) / tiles / 4.0, ms, ms * 1e6 / (tiles * 4.0), ((double)c / (blocks * 16.0)) / (ms * 1e6), h == ref ? "agree" : "DIFFER");
}
int main() {
    std::vector<uint16_t> h(TILES * 128 * 8);
    uint32_t s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; _Float16 f = (_Float16)(((float)(s >> 8) / 16777216.0f - 0.5f) * 4.0f); __builtin_memcpy(&v, &f, 2); }
    uint4* SM; uint32_t* out; unsigned long long* cyc;
    CHK(hipMalloc((void**)&SM, h.size() * 2)); CHK(hipMalloc((void**)&out, 256 * 1024 * 4)); CHK(hipMalloc((void**)&cyc, 8));
    CHK(hipMemcpy(SM, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<uint32_t> ref;
    run<0>(SM, out, cyc, ref); run<1>(SM, out, cyc, ref); run<2>(SM, out, cyc, ref); run<3>(SM, out, cyc, ref); run<4>(SM, out, cyc, ref);
    run<0>(SM, out, cyc, ref);
    return 0;
}
