// elementwise.hip — HBM-bound gather / element-wise kernels of the YOLACT forward.
// Compiled with -ffp-contract=off: every float expression is one IEEE op per operator, in the
// order DESIGN.md §Spec fixes, so results are bit-identical to the oracle on equal inputs.
//
// These replace, for the YOLACT path, the QUANTIZE (input), RESIZE_BILINEAR and max-pool style
// ops inside interpreter.invoke() (/root/reference/src/yolact.rs:163;
// data/FRC_model_edgetpu.log:7-19). All loads/stores are 16 bytes per lane (8 f16 channels),
// consecutive lanes on consecutive channel groups of one pixel: fully coalesced NHWC rows.
#include "yh_internal.h"

namespace yh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// u8 RGB -> f16 (v - mean)/std into the stem's input image: 4 channels (r,g,b,0 = 8 bytes per
// pixel) inside a zero border of 3 pixels (left/top) and >= 5 (right/bottom), so the 7x7 stride-2
// stem needs no padding logic at all and reads 2 pixels per 16-byte chunk.
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void preprocess_rgb8_f16(const uint8_t* __restrict__ rgb, half_t* __restrict__ out,
                                                           int n, int S, int Hp, int Wp) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;   // grid = (ceil(S/256), S, n)
    if (x >= S) return;
    const long long i = ((long long)b * S + y) * S + x;
    const float mean[3] = { 123.68f, 116.78f, 103.94f }, sd[3] = { 58.40f, 57.12f, 57.38f };
    half4v o;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (half_t)(((float)rgb[i * 3 + c] - mean[c]) / sd[c]);
    o[3] = (half_t)0.0f;
    *(half4v*)(out + (((long long)b * Hp + y + 3) * Wp + x + 3) * 4) = o;
}

// grid = (ceil(wo * c8 / 256), ho, n): the only per-lane division is by the compile-unknown c8.
__global__ __launch_bounds__(256) void maxpool3x3s2_f16(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                        int h, int w, int c8, int ho, int wo) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wo * c8) return;
    const int ox = t / c8, cg = t - ox * c8, oy = blockIdx.y, b = blockIdx.z;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if ((unsigned)ix >= (unsigned)w) continue;
            const half8 v = *(const half8*)(x + ((((long long)b * h + iy) * w + ix) * c8 + cg) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = (float)v[e] > m[e] ? (float)v[e] : m[e];
        }
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)m[e];
    *(half8*)(y + ((((long long)b * ho + oy) * wo + ox) * c8 + cg) * 8) = o;
}

// Bilinear resize, align_corners = false: src = (dst + 0.5) * in/out - 0.5 clamped at 0.
// grid = (ceil(wo * c8 / 256), ceil(ho / BIL_ROWS), n): a lane produces BIL_ROWS output rows of one 8-channel group
// (their 4 * BIL_ROWS source loads are all in flight before the first use: the kernel is a 624 MB write stream at the
// protonet's x2 upsample and one 16-byte store per lane left the memory pipeline mostly idle).
#define BIL_ROWS 4
__global__ __launch_bounds__(256) void bilinear_f16(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                    int h, int w, int c8, int ho, int wo,
                                                    long long x_img_stride, long long y_img_stride, uint8_t* __restrict__ y8, const float* __restrict__ y8_inv) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wo * c8) return;
    const int ox = t / c8, cg = t - ox * c8, b = blockIdx.z, oy0 = blockIdx.y * BIL_ROWS;
    const float sy = (float)h / (float)ho, sx = (float)w / (float)wo;
    float fx = ((float)ox + 0.5f) * sx - 0.5f;
    fx = fx < 0.0f ? 0.0f : fx;
    const int x0 = (int)fx, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float lx = fx - (float)x0, hx = 1.0f - lx;
    const half_t* xb = x + b * x_img_stride + cg * 8;
    const int c = c8 * 8;
    half8 p00[BIL_ROWS], p01[BIL_ROWS], p10[BIL_ROWS], p11[BIL_ROWS];
    float ly[BIL_ROWS];
    float inv8[8];   // fp8 output: this lane's eight reciprocal channel scales, loaded once (not behind every row's store)
#pragma unroll
    for (int e = 0; e < 8; ++e) inv8[e] = y8 ? y8_inv[cg * 8 + e] : 1.0f;
#pragma unroll
    for (int r = 0; r < BIL_ROWS; ++r) {
        const int oy = oy0 + r < ho ? oy0 + r : ho - 1;   // (rows past the end recompute the last row and are not stored)
        float fy = ((float)oy + 0.5f) * sy - 0.5f;
        fy = fy < 0.0f ? 0.0f : fy;
        const int y0 = (int)fy, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        ly[r] = fy - (float)y0;
        p00[r] = *(const half8*)(xb + ((long long)y0 * w + x0) * c);
        p01[r] = *(const half8*)(xb + ((long long)y0 * w + x1) * c);
        p10[r] = *(const half8*)(xb + ((long long)y1 * w + x0) * c);
        p11[r] = *(const half8*)(xb + ((long long)y1 * w + x1) * c);
    }
#pragma unroll
    for (int r = 0; r < BIL_ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= ho) break;
        const float hy = 1.0f - ly[r];
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float top = hx * (float)p00[r][e] + lx * (float)p01[r][e];
            const float bot = hx * (float)p10[r][e] + lx * (float)p11[r][e];
            o[e] = (half_t)(hy * top + ly[r] * bot);
        }
        const long long yo = b * y_img_stride + (((long long)oy * wo + ox) * c8 + cg) * 8;
        if (y) *(half8*)(y + yo) = o;
        if (y8) {   // fp8 precision: the consumer is an fp8 convolution (quantised from the f16-rounded value, one scale per channel)
            const unsigned lo = e4m3_pack4((float)o[0] * inv8[0], (float)o[1] * inv8[1], (float)o[2] * inv8[2], (float)o[3] * inv8[3]);
            const unsigned hi = e4m3_pack4((float)o[4] * inv8[4], (float)o[5] * inv8[5], (float)o[6] * inv8[6], (float)o[7] * inv8[7]);
            *(uint2*)(y8 + yo) = make_uint2(lo, hi);
        }
    }
}

// The exact x2 case (ho = 2 h, wo = 2 w: the protonet's upsample). bilinear_f16 above fetches four source pixels per output - 2.5 GB
// through L2 -> CU for a 624 MB tensor at batch 64, which is what bounds it (the vector-memory path of the CU, not HBM). At x2 the
// outputs (2 j + 1, 2 j + 2) x (4 i .. 4 i + 3) read source columns j, j + 1 and rows 2 i - 1 .. 2 i + 2 only: a lane makes those 4 x 2
// outputs of one 8-channel group from 8 fetches (one per output instead of four). Column slots: 0 = output column 0 alone,
// 1 .. w - 1 = the pairs, w = output column 2 w - 1 alone; the two single columns and the first / last block of rows (where the
// generic formula clamps) take the generic per-output path. Every output is the SAME expression of the same four source values
// and the same weights as in bilinear_f16 (weights from the same float formula): bit-identical.
__device__ __forceinline__ void bil_emit(const half8 p00, const half8 p01, const half8 p10, const half8 p11, float lx, float ly,
                                         half_t* __restrict__ y, uint8_t* __restrict__ y8, const float* inv8, long long yo) {
    const float hx = 1.0f - lx, hy = 1.0f - ly;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float top = hx * (float)p00[e] + lx * (float)p01[e];
        const float bot = hx * (float)p10[e] + lx * (float)p11[e];
        o[e] = (half_t)(hy * top + ly * bot);
    }
    if (y) *(half8*)(y + yo) = o;
    if (y8) {
        const unsigned lo = e4m3_pack4((float)o[0] * inv8[0], (float)o[1] * inv8[1], (float)o[2] * inv8[2], (float)o[3] * inv8[3]);
        const unsigned hi = e4m3_pack4((float)o[4] * inv8[4], (float)o[5] * inv8[5], (float)o[6] * inv8[6], (float)o[7] * inv8[7]);
        *(uint2*)(y8 + yo) = make_uint2(lo, hi);
    }
}
__global__ __launch_bounds__(256) void bilinear2x_f16(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                      int h, int w, int c8, long long x_img_stride, long long y_img_stride,
                                                      uint8_t* __restrict__ y8, const float* __restrict__ y8_inv) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (w + 1) * c8) return;
    const int ho = 2 * h, wo = 2 * w;
    const int jp = t / c8, cg = t - jp * c8, b = blockIdx.z, oy0 = blockIdx.y * BIL_ROWS;
    const half_t* xb = x + b * x_img_stride + cg * 8;
    const int c = c8 * 8;
    float inv8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) inv8[e] = y8 ? y8_inv[cg * 8 + e] : 1.0f;
    const int ox_first = jp == 0 ? 0 : 2 * jp - 1, ncol = (jp == 0 || jp == w) ? 1 : 2;
    const int i2 = oy0 / 2;   // source rows i2 - 1 .. i2 + 2
    const bool interior = ncol == 2 && i2 >= 1 && i2 + 2 <= h - 1;   // (oy0 + 3 <= ho - 1 follows; blockIdx.y is wave-uniform, jp nearly)
    if (interior) {
        half8 q[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            q[r][0] = *(const half8*)(xb + ((long long)(i2 - 1 + r) * w + (jp - 1)) * c);
            q[r][1] = *(const half8*)(xb + ((long long)(i2 - 1 + r) * w + jp) * c);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = oy0 + r;
            float fy = ((float)oy + 0.5f) * 0.5f - 0.5f;   // (h / ho = 0.5 exactly; the generic kernel's sy)
            const int y0 = (int)fy;                          // = i2 - 1 + (r + 1) / 2, y1 = y0 + 1: no clamp in an interior block
            const float ly = fy - (float)y0;
            const int a = (r + 1) >> 1;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ox = ox_first + cc;
                const float fx = ((float)ox + 0.5f) * 0.5f - 0.5f;   // = jp - 1 + 0.25 / 0.75: x0 = jp - 1, x1 = jp
                const float lx = fx - (float)(jp - 1);
                bil_emit(q[a][0], q[a][1], q[a + 1][0], q[a + 1][1], lx, ly, y, y8, inv8, b * y_img_stride + (((long long)oy * wo + ox) * c8 + cg) * 8);
            }
        }
        return;
    }
    const float sy = (float)h / (float)ho, sx = (float)w / (float)wo;
    for (int r = 0; r < BIL_ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= ho) break;
        float fy = ((float)oy + 0.5f) * sy - 0.5f;
        fy = fy < 0.0f ? 0.0f : fy;
        const int y0 = (int)fy, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        const float ly = fy - (float)y0;
        for (int cc = 0; cc < ncol; ++cc) {
            const int ox = ox_first + cc;
            float fx = ((float)ox + 0.5f) * sx - 0.5f;
            fx = fx < 0.0f ? 0.0f : fx;
            const int x0 = (int)fx, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
            const float lx = fx - (float)x0;
            bil_emit(*(const half8*)(xb + ((long long)y0 * w + x0) * c), *(const half8*)(xb + ((long long)y0 * w + x1) * c),
                     *(const half8*)(xb + ((long long)y1 * w + x0) * c), *(const half8*)(xb + ((long long)y1 * w + x1) * c), lx, ly, y, y8, inv8,
                     b * y_img_stride + (((long long)oy * wo + ox) * c8 + cg) * 8);
        }
    }
}

// Output reads (not on the hot path): split the fused head rows into loc / conf / mask as f32.
__global__ __launch_bounds__(256) void split_heads_f32(const half_t* __restrict__ heads, long long rows, int ldh,
                                                       int C, float* loc, float* conf, float* mask) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per = 12 + 3 * C + 96;
    if (t >= rows * per) return;
    const long long row = t / per;
    const int j = (int)(t % per);
    const float v = (float)heads[row * ldh + j];
    if (j < 12) { if (loc) loc[row * 12 + j] = v; }
    else if (j < 12 + 3 * C) { if (conf) conf[row * 3 * C + (j - 12)] = v; }
    else if (mask) mask[row * 96 + (j - 12 - 3 * C)] = v;
}

__global__ __launch_bounds__(256) void f16_to_f32(const half_t* __restrict__ x, float* __restrict__ y, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) y[t] = (float)x[t];
}

// Output 4 ("cells"): class logits of anchor 0 on the stride-8 level, [n][cells_l0][C] f32.
__global__ __launch_bounds__(256) void cells_f32(const half_t* __restrict__ heads, int n, int cells_img, int cells_l0,
                                                 int ldh, int C, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * cells_l0 * C) return;
    const int c = (int)(t % C);
    const long long r = t / C;
    const int cell = (int)(r % cells_l0), b = (int)(r / cells_l0);
    out[t] = (float)heads[((long long)b * cells_img + cell) * ldh + 12 + c];
}

// f16 -> OCP FP8 E4M3 codes of x * inv_scale: round to nearest even, saturating at +-448 (groundwork for the
// fp8 convolution path; semantics pinned by oracle/orc_fp8.c). The hardware conversion
// (v_cvt_pk_fp8_f32) does the rounding; the clamp in front makes overflow saturate instead of turning
// into NaN, and NaN inputs keep their sign with the NaN code.
__global__ __launch_bounds__(256) void quantize_e4m3_f16(const half_t* __restrict__ x, uint8_t* __restrict__ y, long long n, float inv_scale) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float v = (float)x[t] * inv_scale;
    uint8_t code;
    code = (uint8_t)e4m3_code(v);
    y[t] = code;
}

// weight panel rows -> E4M3 with one scale per row (output channel), the input tensor's per-channel activation scales folded into
// the K axis first (K index = tap * C + c): y[r][k] = e4m3(((float)x[r][k] * col_scale[k % C]) * inv_scale[r])
__global__ __launch_bounds__(256) void quantize_rows_e4m3_f16(const half_t* __restrict__ x, uint8_t* __restrict__ y, int rows, int K, int C,
                                                              const float* __restrict__ col_scale, const float* __restrict__ inv_scale) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)rows * K) return;
    const int k = (int)(t % K);
    const float v = __fmul_rn((float)x[t], col_scale[k % C]);
    y[t] = (uint8_t)e4m3_code(__fmul_rn(v, inv_scale[t / K]));
}
// ... and the row maxima max_k |(float)x[r][k] * col_scale[k % C]| that its row scales come from (one workgroup per row; bit
// pattern of a non-negative float)
__global__ __launch_bounds__(256) void rowmax_scaled_f16(const half_t* __restrict__ x, int K, int C, const float* __restrict__ col_scale, unsigned* __restrict__ out) {
    __shared__ float red[4];
    const half_t* row = x + (long long)blockIdx.x * K;
    float m = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) { const float a = fabsf(__fmul_rn((float)row[k], col_scale[k % C])); m = a > m ? a : m; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { m = red[0]; for (int i = 1; i < 4; ++i) m = red[i] > m ? red[i] : m; out[blockIdx.x] = __float_as_uint(m); }
}
// max |x| per CHANNEL of an f16 tensor [rows][C] (C a multiple of 8, 256 % (C / 8) == 0), combined with atomicMax on bit patterns
__global__ __launch_bounds__(256) void absmax_channels_f16(const half_t* __restrict__ x, long long rows, int C, unsigned* __restrict__ out) {
    const int c8 = C >> 3, cg = threadIdx.x % c8, rl = threadIdx.x / c8, rpb = 256 / c8;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = 0.0f;
    for (long long r = (long long)blockIdx.x * rpb + rl; r < rows; r += (long long)gridDim.x * rpb) {
        const half8 v = *(const half8*)(x + r * C + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float a = fabsf((float)v[e]); m[e] = a > m[e] ? a : m[e]; }   // (NaN never wins)
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicMax(out + cg * 8 + e, __float_as_uint(m[e]));
}

// max |x| over an f16 range, as the bit pattern of a non-negative float (monotone as unsigned): one atomicMax per wave
__global__ __launch_bounds__(256) void absmax_f16(const half_t* __restrict__ x, long long n8, unsigned* __restrict__ out) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const half8 v = *(const half8*)(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float a = fabsf((float)v[e]); m = a > m ? a : m; }   // (NaN never wins)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void dequant_e4m3_f32(const uint8_t* __restrict__ x, float* __restrict__ y, long long n, const float* __restrict__ scale_ch, int C) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned b = x[t], s = b >> 7, e = (b >> 3) & 15u, m = b & 7u;
    float v;
    if (e == 15u && m == 7u) v = __uint_as_float(0x7FC00000u);
    else if (e == 0u) v = (float)m * 0.001953125f;                          // m / 8 * 2^-6
    else v = __uint_as_float(((e + 120u) << 23) | (m << 20));               // (1 + m / 8) * 2^(e - 7)
    y[t] = (s ? -v : v) * scale_ch[t % C];
}

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_preprocess(const uint8_t* rgb, half_t* out4, int n, int S, int Hp, int Wp, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_rgb8_f16, dim3(nblk(S), (unsigned)S, (unsigned)n), dim3(256), 0, s, rgb, out4, n, S, Hp, Wp);
    return hipGetLastError();
}
hipError_t launch_maxpool3x3s2(const half_t* x, half_t* y, int n, int h, int w, int c, int ho, int wo, hipStream_t s) {
    hipLaunchKernelGGL(maxpool3x3s2_f16, dim3(nblk((long long)wo * (c / 8)), (unsigned)ho, (unsigned)n), dim3(256), 0, s, x, y, h, w, c / 8, ho, wo);
    return hipGetLastError();
}
hipError_t launch_bilinear(const half_t* x, half_t* y, int n, int h, int w, int c, int ho, int wo,
                           long long xs, long long ys, hipStream_t s, uint8_t* y8, const float* y8_inv) {
    if (ho == 2 * h && wo == 2 * w && w >= 2)
        hipLaunchKernelGGL(bilinear2x_f16, dim3(nblk((long long)(w + 1) * (c / 8)), (unsigned)((ho + BIL_ROWS - 1) / BIL_ROWS), (unsigned)n), dim3(256), 0, s, x, y, h, w, c / 8, xs, ys, y8, y8_inv);
    else
        hipLaunchKernelGGL(bilinear_f16, dim3(nblk((long long)wo * (c / 8)), (unsigned)((ho + BIL_ROWS - 1) / BIL_ROWS), (unsigned)n), dim3(256), 0, s, x, y, h, w, c / 8, ho, wo, xs, ys, y8, y8_inv);
    return hipGetLastError();
}
hipError_t launch_quantize_rows_e4m3(const half_t* x, uint8_t* y, int rows, int K, int C, const float* col_scale, const float* inv_scale_rows, hipStream_t s) {
    hipLaunchKernelGGL(quantize_rows_e4m3_f16, dim3(nblk((long long)rows * K)), dim3(256), 0, s, x, y, rows, K, C, col_scale, inv_scale_rows);
    return hipGetLastError();
}
hipError_t launch_rowmax_scaled_f16(const half_t* x, int rows, int K, int C, const float* col_scale, unsigned* out_bits, hipStream_t s) {
    hipLaunchKernelGGL(rowmax_scaled_f16, dim3((unsigned)rows), dim3(256), 0, s, x, K, C, col_scale, out_bits);
    return hipGetLastError();
}
hipError_t launch_absmax_channels_f16(const half_t* x, long long rows, int C, unsigned* out_bits, hipStream_t s) {
    if (C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return hipErrorInvalidValue;
    const long long rpb = 256 / (C / 8), want = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(absmax_channels_f16, dim3((unsigned)(want < 1024 ? (want > 0 ? want : 1) : 1024)), dim3(256), 0, s, x, rows, C, out_bits);
    return hipGetLastError();
}
hipError_t launch_absmax_f16(const half_t* x, long long n, unsigned* out_bits, hipStream_t s) {
    if (n % 8 != 0) return hipErrorInvalidValue;
    const long long n8 = n / 8;
    const unsigned grid = (unsigned)(n8 < 256ll * 2048 ? (n8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_f16, dim3(grid ? grid : 1), dim3(256), 0, s, x, n8, out_bits);
    return hipGetLastError();
}
hipError_t launch_dequant_e4m3_f32(const uint8_t* x, float* y, long long n, const float* scale_ch, int C, hipStream_t s) {
    hipLaunchKernelGGL(dequant_e4m3_f32, dim3(nblk(n)), dim3(256), 0, s, x, y, n, scale_ch, C);
    return hipGetLastError();
}
hipError_t launch_quantize_e4m3(const half_t* x, uint8_t* y, long long n, float inv_scale, hipStream_t s) {
    hipLaunchKernelGGL(quantize_e4m3_f16, dim3(nblk(n)), dim3(256), 0, s, x, y, n, inv_scale);
    return hipGetLastError();
}
hipError_t launch_split_heads(const half_t* heads, int n, int cells, int ldh, int C, float* loc, float* conf,
                              float* mask, hipStream_t s) {
    const long long rows = (long long)n * cells;
    hipLaunchKernelGGL(split_heads_f32, dim3(nblk(rows * (12 + 3 * C + 96))), dim3(256), 0, s, heads, rows, ldh, C, loc, conf, mask);
    return hipGetLastError();
}
hipError_t launch_f16_to_f32(const half_t* x, float* y, long long n, hipStream_t s) {
    hipLaunchKernelGGL(f16_to_f32, dim3(nblk(n)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}
hipError_t launch_cells_f32(const half_t* heads, int n, int cells_img, int cells_l0, int ldh, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(cells_f32, dim3(nblk((long long)n * cells_l0 * C)), dim3(256), 0, s, heads, n, cells_img, cells_l0, ldh, C, out);
    return hipGetLastError();
}

__global__ void side_touch(unsigned* w) { *w = 0u; }
hipError_t launch_side_touch(unsigned* w, hipStream_t s) {
    hipLaunchKernelGGL(side_touch, dim3(1), dim3(1), 0, s, w);
    return hipGetLastError();
}

}  // namespace yh
