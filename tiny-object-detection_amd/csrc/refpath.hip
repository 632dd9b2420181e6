// refpath.hip — the reference's own pre/post-processing (SURVEY.md A3, A5, A8, A8', A9) on device.
//
// Device restatement of /root/reference/src/yolact.rs:
//   :195-201,:134-140  u32 -> RGB bytes (to_be_bytes()[..3])        -> fused into resize_v_u32
//   :208, :231         image::resize_exact(.., Triangle)            -> resize_v_* + resize_h
//   :213-214           crop two S x S tiles                         -> resize_h mode 1
//   :108-118           gated argmax over the first 4 of C logits    -> cells_postprocess
//   :52-88             terrible_id flood fill                       -> cells_postprocess (strict/sane)
//   :127-128           pack `cls<<24 & id<<16`, x8 nearest upsample -> cells_postprocess + upsample_codes
//   :219-220           stitch t1|t2                                 -> upsample_codes (stitched)
//   :230-233           repack u32                                   -> resize_h mode 2
// Integer logic is exact; the resampler follows image 0.24.1's separable f32 algorithm with one
// explicit round-to-nearest op per operator, bit-identical to oracle/orc_ref.c.
// HBM-bound byte work: one lane per output pixel, consecutive lanes on consecutive pixels.
#include "yh_internal.h"

namespace yh {

struct SampleWin { int left, n; float inputc, sratio, sum; };

__device__ __forceinline__ float tri_w(int i, float inputc, float sratio) {
    const float x = __fdiv_rn(__fsub_rn((float)i, inputc), sratio);
    const float a = fabsf(x);
    return a < 1.0f ? __fsub_rn(1.0f, a) : 0.0f;
}

__device__ __forceinline__ SampleWin sample_window(int out_i, int in_size, int out_size) {
    SampleWin s;
    const float ratio = __fdiv_rn((float)in_size, (float)out_size);
    s.sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = __fmul_rn(1.0f, s.sratio);
    float inputc = __fmul_rn(__fadd_rn((float)out_i, 0.5f), ratio);
    long long left = (long long)floorf(__fsub_rn(inputc, support));
    left = left < 0 ? 0 : (left > in_size - 1 ? in_size - 1 : left);
    long long right = (long long)ceilf(__fadd_rn(inputc, support));
    right = right < left + 1 ? left + 1 : (right > in_size ? in_size : right);
    s.inputc = __fsub_rn(inputc, 0.5f);
    s.left = (int)left;
    s.n = (int)(right - left);
    float sum = 0.0f;
    for (int i = 0; i < s.n; ++i) sum = __fadd_rn(sum, tri_w(s.left + i, s.inputc, s.sratio));
    s.sum = sum;
    return s;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void resize_v(const void* __restrict__ src, int sw, int sh,
                                                float* __restrict__ tmp, int dh) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= dh * sw) return;
    const int oy = t / sw, x = t - oy * sw;
    const SampleWin s = sample_window(oy, sh, dh);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int i = 0; i < s.n; ++i) {
        const float w = __fdiv_rn(tri_w(s.left + i, s.inputc, s.sratio), s.sum);
        const long long idx = (long long)(s.left + i) * sw + x;
        float c0, c1, c2;
        if (PACKED) {
            const uint32_t px = ((const uint32_t*)src)[idx];
            c0 = (float)(px >> 24); c1 = (float)((px >> 16) & 0xFFu); c2 = (float)((px >> 8) & 0xFFu);
        } else {
            const uint8_t* q = (const uint8_t*)src + idx * 3;
            c0 = (float)q[0]; c1 = (float)q[1]; c2 = (float)q[2];
        }
        a0 = __fadd_rn(a0, __fmul_rn(c0, w));
        a1 = __fadd_rn(a1, __fmul_rn(c1, w));
        a2 = __fadd_rn(a2, __fmul_rn(c2, w));
    }
    float* o = tmp + (long long)t * 3;
    o[0] = a0; o[1] = a1; o[2] = a2;
}

__device__ __forceinline__ uint32_t to_u8(float v) {
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    return (uint32_t)roundf(v);
}

__global__ __launch_bounds__(256) void resize_h(const float* __restrict__ tmp, int sw, int dh,
                                                void* __restrict__ dst, int dw, int mode) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= dh * dw) return;
    const int y = t / dw, ox = t - y * dw;
    const SampleWin s = sample_window(ox, sw, dw);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int i = 0; i < s.n; ++i) {
        const float w = __fdiv_rn(tri_w(s.left + i, s.inputc, s.sratio), s.sum);
        const float* q = tmp + ((long long)y * sw + s.left + i) * 3;
        a0 = __fadd_rn(a0, __fmul_rn(q[0], w));
        a1 = __fadd_rn(a1, __fmul_rn(q[1], w));
        a2 = __fadd_rn(a2, __fmul_rn(q[2], w));
    }
    const uint32_t r = to_u8(a0), g = to_u8(a1), b = to_u8(a2);
    if (mode == 2) {
        ((uint32_t*)dst)[t] = (r << 24) | (g << 16) | (b << 8);
    } else {
        long long o;
        if (mode == 1) { const int S = dh, tile = ox / S, tx = ox - tile * S; o = ((long long)tile * S * S + (long long)y * S + tx) * 3; }
        else o = (long long)t * 3;
        uint8_t* d = (uint8_t*)dst + o;
        d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)b;
    }
}

#define YH_GRID_MAX 4096

// One workgroup per tile. cells: [n_tiles][ncells][C] f32.
__global__ __launch_bounds__(256) void cells_postprocess(const float* __restrict__ cells, int grid, int C, int mode,
                                                         uint32_t* __restrict__ codes, int* __restrict__ diverged) {
    __shared__ uint8_t cls[YH_GRID_MAX];
    __shared__ int label[YH_GRID_MAX];
    __shared__ int flag, changed;
    const int tile = blockIdx.x, tid = threadIdx.x, nc = grid * grid;
    const float* base = cells + (long long)tile * nc * C;
    if (tid == 0) flag = 0;
    // yolact.rs:108-118: running max from 0.0, strict '>', first four logits; 4-arm match
    for (int c = tid; c < nc; c += 256) {
        const float* a = base + (long long)c * C;
        float mx = 0.0f;
        bool b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float v = a[i]; b[i] = v > mx; mx = b[i] ? v : mx; }
        uint8_t k;
        if (!b[0] && b[1] && !b[2] && !b[3]) k = 1;
        else if (!b[0] && b[2] && !b[3]) k = 2;
        else if (!b[0] && b[3]) k = 3;
        else k = 0;
        cls[c] = k;
    }
    __syncthreads();
    if (mode == 0) {
        // yolact.rs:52-88: the flood fill terminates iff no class-3 cell has a class-3 neighbour in
        // LINEAR index space (px-1, px+1, px-grid, px+grid, with usize wrap failing img.get), and
        // then labels nothing: every id stays -1 (DESIGN.md §Compat gives the argument).
        for (int c = tid; c < nc; c += 256) {
            if (cls[c] != 3) continue;
            const bool nb = (c >= 1 && cls[c - 1] == 3) || (c + 1 < nc && cls[c + 1] == 3) ||
                            (c >= grid && cls[c - grid] == 3) || (c + grid < nc && cls[c + grid] == 3);
            if (nb) flag = 1;
        }
        __syncthreads();
        // yolact.rs:127: (cls << 24) & ((-1i8 as u32) << 16) == cls << 24
        for (int c = tid; c < nc; c += 256) codes[(long long)tile * nc + c] = ((uint32_t)cls[c] << 24) & (0xFFFFFFFFu << 16);
        if (tid == 0) diverged[tile] = flag;
        return;
    }
    // SANE: 4-connected components, min-index label propagation, ids in raster order of roots
    for (int c = tid; c < nc; c += 256) label[c] = cls[c] == 3 ? c : -1;
    __syncthreads();
    for (int iter = 0; iter < nc; ++iter) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int c = tid; c < nc; c += 256) {
            int l = label[c];
            if (l < 0) continue;
            const int y = c / grid, x = c - y * grid;
            int m = l;
            if (x > 0 && label[c - 1] >= 0) m = min(m, label[c - 1]);
            if (x < grid - 1 && label[c + 1] >= 0) m = min(m, label[c + 1]);
            if (y > 0 && label[c - grid] >= 0) m = min(m, label[c - grid]);
            if (y < grid - 1 && label[c + grid] >= 0) m = min(m, label[c + grid]);
            if (m < l) { label[c] = m; changed = 1; }
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int c = tid; c < nc; c += 256) {
        const int l = label[c];
        uint32_t idb = 0xFFu;
        if (l >= 0) {
            int rank = 0;
            for (int r = 0; r < l; ++r) rank += (label[r] == r) ? 1 : 0;
            idb = (uint32_t)(rank > 127 ? 127 : rank);
        }
        codes[(long long)tile * nc + c] = ((uint32_t)cls[c] << 24) | (idb << 16);
    }
    if (tid == 0) diverged[tile] = 0;
}

__global__ __launch_bounds__(256) void upsample_codes(const uint32_t* __restrict__ codes, int n_tiles, int grid,
                                                      uint32_t* __restrict__ out, int stitched) {
    const int S = grid * 8;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_tiles * S * S) return;
    int tile, y, x;
    if (stitched) { const int W = n_tiles * S; y = (int)(t / W); const int xx = (int)(t - (long long)y * W); tile = xx / S; x = xx - tile * S; }
    else { tile = (int)(t / (S * S)); const int r = (int)(t - (long long)tile * S * S); y = r / S; x = r - y * S; }
    out[t] = codes[(long long)tile * grid * grid + (y >> 3) * grid + (x >> 3)];
}

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_resize_v_u32(const uint32_t* src, int sw, int sh, float* tmp, int dh, hipStream_t s) {
    hipLaunchKernelGGL((resize_v<true>), dim3(nblk((long long)dh * sw)), dim3(256), 0, s, (const void*)src, sw, sh, tmp, dh);
    return hipGetLastError();
}
hipError_t launch_resize_v_rgb8(const uint8_t* src, int sw, int sh, float* tmp, int dh, hipStream_t s) {
    hipLaunchKernelGGL((resize_v<false>), dim3(nblk((long long)dh * sw)), dim3(256), 0, s, (const void*)src, sw, sh, tmp, dh);
    return hipGetLastError();
}
hipError_t launch_resize_h(const float* tmp, int sw, int dh, void* dst, int dw, int mode, hipStream_t s) {
    hipLaunchKernelGGL(resize_h, dim3(nblk((long long)dh * dw)), dim3(256), 0, s, tmp, sw, dh, dst, dw, mode);
    return hipGetLastError();
}
hipError_t launch_cells_postprocess(const float* cells, int n_tiles, int grid, int C, int mode, uint32_t* codes,
                                    int* diverged, hipStream_t s) {
    hipLaunchKernelGGL(cells_postprocess, dim3((unsigned)n_tiles), dim3(256), 0, s, cells, grid, C, mode, codes, diverged);
    return hipGetLastError();
}
hipError_t launch_upsample_codes(const uint32_t* codes, int n_tiles, int grid, uint32_t* out, int stitched, hipStream_t s) {
    const long long S = grid * 8;
    hipLaunchKernelGGL(upsample_codes, dim3(nblk((long long)n_tiles * S * S)), dim3(256), 0, s, codes, n_tiles, grid, out, stitched);
    return hipGetLastError();
}

}  // namespace yh
