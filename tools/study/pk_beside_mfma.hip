// Does packed-f32 arithmetic of one kernel go wrong while another kernel's MFMA-dense waves share the chip?  (round 5: the tail's candidate
// kernel, full of v_pk_fma_f32, gave wrong sums in lanes 48-63 only while proto3's launch - the two-stage 128 x 128 MFMA tile - ran beside
// it)  Stream A: a kernel that does nothing but v_mfma_f32_32x32x16_f16 back to back for a few milliseconds.
// Stream B: a self-checking victim - every lane runs the same chain of fused multiply-adds twice, once as v_pk_fma_f32 on a register pair
// (inline assembly: the packed instruction for certain) and once as two v_fma_f32, and counts chains whose bits differ, by quarter of
// the wave. Usage: pk_beside_mfma.bin [trials = 20] [idle_ms = 2000]     Build: hipcc --offload-arch=gfx950 -O3 -o pk_beside_mfma.bin pk_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void mfma_dense(float* sink, int iters) {
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    float s = 0.0f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) sink[0] = s;   // (never: keeps the loop)
}

// 192 threads = 3 waves, as the candidate kernel; each lane: CH independent chains of LEN packed FMAs, checked against scalar FMAs
template <int CH, int LEN>
__global__ __launch_bounds__(192) void pk_victim(unsigned long long* bad /* [4] by quarter */, unsigned seed) {
    const unsigned t = blockIdx.x * 192 + threadIdx.x;
    unsigned long long wrong = 0;
    for (int ch = 0; ch < CH; ++ch) {
        const float x0 = 1.0f + 1e-3f * (float)((t * 2654435761u + seed + ch * 40503u) >> 20);
        const float x1 = 0.5f + 1e-3f * (float)((t * 40503u + seed * 7u + ch) >> 21);
        const float m0 = 0.99993896484375f, m1 = 1.00006103515625f, k0 = 1e-4f, k1 = -1e-4f;
        f32x2 acc = { x0, x1 };
        const f32x2 mm = { m0, m1 }, kk = { k0, k1 };
        float s0 = x0, s1 = x1;
#pragma unroll
        for (int i = 0; i < LEN; ++i) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(mm), "v"(kk));
            s0 = __builtin_fmaf(s0, m0, k0);
            s1 = __builtin_fmaf(s1, m1, k1);
        }
        wrong += (__float_as_uint(acc[0]) != __float_as_uint(s0)) || (__float_as_uint(acc[1]) != __float_as_uint(s1));
    }
    if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 20, idle_ms = argc > 2 ? atoi(argv[2]) : 2000;
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* bad; hipMalloc(&bad, 32);
    for (int mode = 0; mode < 2; ++mode) {   // 0: the victim alone, 1: beside the MFMA kernel
        unsigned long long tot[4] = {0, 0, 0, 0};
        for (int tr = 0; tr < trials; ++tr) {
            if (idle_ms) usleep(1000 * idle_ms);
            hipMemsetAsync(bad, 0, 32, sb); hipStreamSynchronize(sb);
            if (mode) mfma_dense<<<298, 256, 0, sa>>>(sink, 60000);   // a few ms of back-to-back MFMAs, 298 workgroups as proto3 at batch 1
            for (int l = 0; l < 40; ++l) pk_victim<64, 256><<<101, 192, 0, sb>>>(bad, (unsigned)(tr * 131 + l));
            hipDeviceSynchronize();
            unsigned long long h[4]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
            for (int q = 0; q < 4; ++q) tot[q] += h[q];
        }
        printf("%s: %d trials x 40 launches x 101 x 192 lanes x 64 chains of 256 packed FMAs: wrong chains by wave quarter %llu %llu %llu %llu\n",
               mode ? "beside the MFMA kernel" : "alone", trials, tot[0], tot[1], tot[2], tot[3]);
    }
    return 0;
}
