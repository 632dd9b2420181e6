// dma_probe.hip — how fast does a workgroup pull a weight tile [ROWS][128 B] out of L2 into LDS by LDS-DMA, as a function of the
// matrix's row pitch? NOT part of the product library. Every workgroup of the grid reads the SAME matrix (it stays in L2), tile after
// tile along K, the way the 1x1 convolutions and the bottleneck kernels stream their weight panels.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/dma_probe tools/exp/dma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) char lds_char;

// ROWS x 128 B tile; ld = row pitch in bytes; tiles walk along the row (k) and wrap; depth = tiles in flight (1 or 2 LDS stages)
template <int ROWS, int DEPTH>
__global__ __launch_bounds__(256, 1) void pull(const char* w, unsigned w_bytes, int ld, int ktiles, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[DEPTH * ROWS * 128];
    lds_char* const lds3 = (lds_char*)lds;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), pc = tid & 7, rb = tid >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);
    auto issue = [&](int t) {
        lds_char* const d = lds3 + (t % DEPTH) * (ROWS * 128) + wave * 1024;
        const unsigned col = (unsigned)((t % ktiles) * 128 + pc * 16);
#pragma unroll
        for (int p = 0; p < ROWS / 32; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d + p * 4096, 16, (int)((unsigned)((rb + 32 * p) * ld) + col), 0, 0, 0);
    };
    float acc = 0.0f;
    if (DEPTH == 2) issue(0);
    for (int t = 0; t < iters; ++t) {
        if (DEPTH == 2) {
            if (t + 1 < iters) { issue(t + 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS / 32) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else { issue(t); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        acc += *(const float*)(lds + (t % DEPTH) * (ROWS * 128) + tid * 16);   // (one read per tile so that nothing is optimised away)
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.0f) sink[0] = acc;
}

int main() {
    const int rows = 256, iters = 256;
    char* w; float* sink;
    const size_t bytes = (size_t)rows * 8192 + 65536;
    CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0, bytes)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    auto run = [&](const char* name, int ld, int ktiles, auto kern, int tile_bytes) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 0, 0, w, (unsigned)bytes, ld, ktiles, iters, sink);
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 0, 0, w, (unsigned)bytes, ld, ktiles, iters, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        const double per_cu = (double)iters * tile_bytes / (ms * 1e-3);
        printf("%-64s %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, ms, per_cu / 1e9, per_cu * cus / 1e12);
    };
    printf("%d CUs, one workgroup each, %d tiles per workgroup\n", cus, iters);
    run("256 rows x 128 B, pitch 128 B (tile contiguous), 1 stage", 128, 1, pull<256, 1>, 32768);
    run("256 rows x 128 B, pitch 512 B (K = 256), 1 stage", 512, 4, pull<256, 1>, 32768);
    run("256 rows x 128 B, pitch 2048 B (K = 1024), 1 stage", 2048, 16, pull<256, 1>, 32768);
    run("256 rows x 128 B, pitch 4096 B (K = 2048), 1 stage", 4096, 32, pull<256, 1>, 32768);
    run("256 rows x 128 B, pitch 128 B (tile contiguous), 2 stages", 128, 1, pull<256, 2>, 32768);
    run("256 rows x 128 B, pitch 512 B (K = 256), 2 stages", 512, 4, pull<256, 2>, 32768);
    run("256 rows x 128 B, pitch 2048 B (K = 1024), 2 stages", 2048, 16, pull<256, 2>, 32768);
    run("256 rows x 128 B, pitch 4096 B (K = 2048), 2 stages", 4096, 32, pull<256, 2>, 32768);
    run("256 rows x 128 B, pitch 2176 B (K = 1024 + 64 pad), 2 stages", 2176, 16, pull<256, 2>, 32768);
    run("64 rows x 128 B, pitch 512 B, 2 stages", 512, 4, pull<64, 2>, 8192);
    run("64 rows x 128 B, pitch 2048 B, 2 stages", 2048, 16, pull<64, 2>, 8192);
    return 0;
}
