// stream_probe.hip — what does this box's HBM give a kernel shaped like the epilogue-bound 1x1 expand convs?
// NOT part of the product library. Each variant moves the expand conv's bytes (read a 1/4-wide input, read a
// residual, write an output) with no arithmetic to speak of, so its rate is the ceiling those launches can be held against.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/exp/stream_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// y = r + x[i / 4] : reads n/4 + n, writes n (16 bytes per lane, one element per thread)
__global__ __launch_bounds__(256) void add_oneshot(const half8* __restrict__ x, const half8* __restrict__ r, half8* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = r[i] + x[i >> 2];
}
// the same, U elements per thread (all loads first)
template <int U>
__global__ __launch_bounds__(256) void add_unrolled(const half8* __restrict__ x, const half8* __restrict__ r, half8* __restrict__ y, long long n) {
    const long long base = ((long long)blockIdx.x * U) * 256 + threadIdx.x;
    half8 rv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = base + u * 256; if (i < n) { rv[u] = r[i]; xv[u] = x[i >> 2]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = base + u * 256; if (i < n) y[i] = rv[u] + xv[u]; }
}
__global__ __launch_bounds__(256) void copy_oneshot(const half8* __restrict__ r, half8* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = r[i];
}
__global__ __launch_bounds__(256) void read_only(const half8* __restrict__ r, half8* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    half8 v = r[i < n ? i : 0];
    if (v[0] == (_Float16)12345.0f) y[0] = v;   // (never)
}
__global__ __launch_bounds__(256) void write_only(half8* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    half8 v; for (int e = 0; e < 8; ++e) v[e] = (_Float16)1.0f;
    if (i < n) y[i] = v;
}

// the conv epilogue's pattern: workgroup t owns rows [TMR (t / NCT), + TMR) x 16-byte chunks [CW (t % NCT), + CW) of an [M][C] matrix
// (CW chunks = one channel tile), thread = one chunk of one row per pass; y = r + x (x: [M][C / 4], its matching quarter row)
template <int TMR, int CW, int DEPTH>
__global__ __launch_bounds__(256) void add_tiled(const half8* __restrict__ x, const half8* __restrict__ r, half8* __restrict__ y, int M, int C8) {
    const int nct = C8 / CW, t = blockIdx.x, ct = t % nct, mt = t / nct;
    constexpr int RPP = 256 / CW, NP = TMR / RPP;
    const int c = ct * CW + threadIdx.x % CW, r0 = mt * TMR + threadIdx.x / CW;
    half8 rv[NP], xv[NP];
#pragma unroll
    for (int q = 0; q < NP; q += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { const int m = r0 + (q + d) * RPP; if (m < M) { rv[q + d] = r[(long long)m * C8 + c]; xv[q + d] = x[(long long)m * (C8 / 4) + (c >> 2)]; } }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { const int m = r0 + (q + d) * RPP; if (m < M) y[(long long)m * C8 + c] = rv[q + d] + xv[q + d]; }
    }
}

int main() {
    const long long M = 64LL * 35 * 35, C = 1024;          // layer 3 at batch 64
    const long long n = M * C / 8;                          // half8 elements of the output
    half8 *x, *r, *y;
    CK(hipMalloc(&x, n / 4 * 16)); CK(hipMalloc(&r, n * 16)); CK(hipMalloc(&y, n * 16));
    CK(hipMemset(x, 0, n / 4 * 16)); CK(hipMemset(r, 0, n * 16)); CK(hipMemset(y, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    const unsigned g1 = (unsigned)((n + 255) / 256);
    const double B = (double)n * 16;
    printf("output %.1f MB (64 x 35 x 35 x 1024 f16)\n", B / 1e6);
    time("read-only (r)", B, [&] { hipLaunchKernelGGL(read_only, dim3(g1), dim3(256), 0, 0, r, y, n); });
    time("write-only (y)", B, [&] { hipLaunchKernelGGL(write_only, dim3(g1), dim3(256), 0, 0, y, n); });
    time("copy y = r", 2 * B, [&] { hipLaunchKernelGGL(copy_oneshot, dim3(g1), dim3(256), 0, 0, r, y, n); });
    time("hipMemcpyDtoD y = r", 2 * B, [&] { CK(hipMemcpyAsync(y, r, n * 16, hipMemcpyDeviceToDevice, 0)); });
    time("y = r + x[i/4], 1 per thread", 2.25 * B, [&] { hipLaunchKernelGGL(add_oneshot, dim3(g1), dim3(256), 0, 0, x, r, y, n); });
    time("y = r + x[i/4], 4 per thread", 2.25 * B, [&] { hipLaunchKernelGGL(add_unrolled<4>, dim3((g1 + 3) / 4), dim3(256), 0, 0, x, r, y, n); });
    time("y = r + x[i/4], 8 per thread", 2.25 * B, [&] { hipLaunchKernelGGL(add_unrolled<8>, dim3((g1 + 7) / 8), dim3(256), 0, 0, x, r, y, n); });
    const int Mi = (int)M, C8 = (int)(C / 8);
    auto tiled = [&](const char* name, auto kern, int tmr, int cw) {
        const unsigned g = (unsigned)(((Mi + tmr - 1) / tmr) * (C8 / cw));
        time(name, 2.25 * B, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, 0, x, r, y, Mi, C8); });
    };
    tiled("tile 128 rows x 256 B, all loads first", add_tiled<128, 16, 8>, 128, 16);
    tiled("tile 128 rows x 256 B, 2 passes at a time", add_tiled<128, 16, 2>, 128, 16);
    tiled("tile 128 rows x 256 B, 1 pass at a time", add_tiled<128, 16, 1>, 128, 16);
    tiled("tile 256 rows x 128 B, all loads first", add_tiled<256, 8, 8>, 256, 8);
    tiled("tile 256 rows x 128 B, 1 pass at a time", add_tiled<256, 8, 1>, 256, 8);
    tiled("tile 32 rows x 2048 B (whole rows), all first", add_tiled<32, 128, 16>, 32, 128);
    tiled("tile 64 rows x 1024 B, all first", add_tiled<64, 64, 16>, 64, 64);
    return 0;
}
