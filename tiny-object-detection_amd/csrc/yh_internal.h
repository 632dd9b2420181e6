// yh_internal.h — shared declarations of libyolact_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/yolact_hip_debug.h"   // (the public header + the named tuning fields, measurement / test entry points)

namespace yh {

typedef _Float16 half_t;

// ---------------------------------------------------------------------------------------------
// Tracing: roctx ranges around set-input / forward / tail / read-back and the TFLite op groups, so that a
// `rocprofv3 --marker-trace --kernel-trace -- python3 bench.py` timeline reads without symbol archaeology. The two entry
// points are looked up in the PROCESS at first use (dlsym(RTLD_DEFAULT)): rocprofv3 --marker-trace preloads
// librocprofiler-sdk-roctx.so and a host may link it; when neither did, a range costs one null test. No dependency.
// ---------------------------------------------------------------------------------------------
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const RoctxApi& roctx_api();   // engine.hip
struct TraceRange {
    bool on;
    explicit TraceRange(const char* name) { const RoctxApi& r = roctx_api(); on = r.push && r.pop; if (on) r.push(name); }
    ~TraceRange() { if (on) roctx_api().pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

// ---------------------------------------------------------------------------------------------
// Setup audit (engine.hip): once-only work - graph captures - is done on the calling thread, and a group
// does it for all its members one after the other BEFORE its worker threads run. SetupScope brackets such a section, WorkerScope
// a worker thread's job; an overlap of the two anywhere in the process is counted and yh_debug_setup_audit reports it (the
// tests assert 0). Counters only: nothing waits on them.
// ---------------------------------------------------------------------------------------------
struct SetupAudit { std::atomic<int> setup_active{0}, worker_active{0}; std::atomic<long long> setups{0}, worker_jobs{0}, overlaps{0}; };
SetupAudit& setup_audit();   // engine.hip
struct SetupScope {
    SetupScope() { SetupAudit& a = setup_audit(); a.setups++; a.setup_active++; if (a.worker_active.load() > 0) a.overlaps++; }
    ~SetupScope() { SetupAudit& a = setup_audit(); if (a.worker_active.load() > 0) a.overlaps++; a.setup_active--; }
    SetupScope(const SetupScope&) = delete;
    SetupScope& operator=(const SetupScope&) = delete;
};
struct WorkerScope {
    WorkerScope() { SetupAudit& a = setup_audit(); a.worker_jobs++; a.worker_active++; if (a.setup_active.load() > 0) a.overlaps++; }
    ~WorkerScope() { setup_audit().worker_active--; }
    WorkerScope(const WorkerScope&) = delete;
    WorkerScope& operator=(const WorkerScope&) = delete;
};
// group.hip drives its members through these (not part of the C ABI)
void engine_set_worker_mode(yh_engine* h, bool on);
bool engine_step_prepared(const yh_engine* h, int n_frames, int with_tail);   // eager handles: always true
bool engine_uses_graph(const yh_engine* h);
bool rccl_shared_device_allowed();   // yh_debug_rccl_shared_device (tests with a stand-in librccl: several ranks on one device)

// ---------------------------------------------------------------------------------------------
// Convolution as implicit GEMM (conv_igemm.hip).
//   D[ch][m] = sum_k Wt[ch][k] * X[m][k],  m = (n, p, q) row-major,  k = ((r*S + s)*C + c).
// Activations are NHWC f16 with C a multiple of 8 (C >= 64: multiple of 64; C == 8: "small-C"
// mode, one (r,s) tap per 16-byte chunk, taps listed in rs_table). Weights are a zero-padded
// [coutPad][Kpad] f16 panel, Kpad a multiple of 64, coutPad a multiple of the channel tile.
// ---------------------------------------------------------------------------------------------
struct ConvParams {
    const half_t* x;      // input base (first image, first pixel of the addressed slice)
    const half_t* w;      // weight panel [coutPad][ldw]
    const float* bias;    // [coutPad] (zero padded)
    const half_t* res;    // residual or nullptr
    // res_up: the residual is the bilinear resize (align_corners = false) of the lower-resolution tensor res
    // [n][res_h][res_w][ldres] to this conv's P x Q output, computed in the epilogue and rounded to f16 first - the FPN
    // top-down add without materialising the upsampled tensor (bit-identical to bilinear_f16 + a plain residual)
    int res_up, res_h, res_w;
    half_t* y;            // output base
    const int2* rs_table; // small-C mode: (r, s) per 16-byte chunk index; r = 1<<20 marks padding
    long long x_img_stride, y_img_stride, res_img_stride; // elements per image
    unsigned x_bytes;     // bytes addressable from x (buffer descriptor range, includes the zero block)
    unsigned x_zero_off;  // byte offset from x of a 16-byte block of zeros (padded taps read it)
    unsigned w_bytes;     // bytes of the weight panel
    int N, H, W, C;       // input
    int P, Q;             // output spatial
    int R, S, stride, pad;
    int M;                // N*P*Q (rows this launch may write: m < M)
    int m_tile0;          // first row tile of this launch (two-phase launches: the tail starts past 0)
    int ch_tile0;         // first channel tile of this launch (channel-split launches), in units of this launch's tile
    int ch_base;          // ... plus this many channels (a remainder launch whose first channel is no multiple of its tile: 256 on 96-channel tiles)
    // multi-level input (nlev > 0): the P*Q rows of an image are the cells of nlev pyramid levels laid end to
    // end (level l: rows lev_start[l].., a lev_h[l] x lev_w[l] image); taps stay inside their level
    int nlev, lev_start[5], lev_h[5], lev_w[5];
    // fp8 form (experimental): x and w hold OCP E4M3 bytes; C, ldw and the image strides are in
    // 2-byte units (two fp8 values), so the loader is the f16 one; out = acc * scale[ch] + bias
    const float* scale;
    // E4M3 output (fp8 precision: this tensor feeds an fp8 convolution): y8[same element offsets as y] =
    // e4m3((float)(f16 result) * y8_inv[channel]), round to nearest even, saturating - one (reciprocal) activation scale per
    // channel of the tensor, [ldy] floats (round 4; a per-tensor scale is the same value everywhere). y may then be nullptr.
    uint8_t* y8;
    const float* y8_inv;
    // two-source form (x2 != nullptr; 1x1 convolutions only): K is the concatenation of x's C channels (k-steps
    // 0 .. k1steps-1) and of C2 channels of a SECOND tensor x2 [n][H2][W2][C2] read at stride2 (output pixel (p, q) <-
    // x2 pixel (p*stride2, q*stride2)); the weight panel holds [W1 row | W2 row] along K. A bottleneck block's last 1x1
    // conv and its 1x1 downsample projection as ONE accumulation: the projection is never written or re-read.
    const half_t* x2;
    long long x2_img_stride;
    unsigned x2_bytes, x2_zero_off;
    int W2, C2, stride2, k1steps;
    // fused 1x1 tail (w2 != nullptr; 256-channel tiles that hold ALL of this conv's output channels): the tile's rounded
    // f16 outputs stay in LDS and a second convolution y2[m][0..31] = relu(bias2 + sum_c w2[o][c] * y[m][c]) runs on them
    // in the epilogue (protonet: the last 3x3 conv and the 1x1 that makes the 32 prototypes); y may then be nullptr.
    const half_t* w2;     // [32][256] f16
    const float* bias2;   // [32]
    half_t* y2;           // [M][32] dense
    int skip_dma;         // timing only (tune.ablate bit 2): the loader issues NO LDS-DMA instruction (results are garbage)
    int cout8;            // output channels rounded up to 8 (stores happen in 8-channel chunks)
    int ldw;              // Kpad
    int ksteps;           // Kpad / 64
    int ldy, ldres;       // row strides in elements
    int y_dense;          // 1: y offset = m*ldy (no per-row division)
    int act;              // 0 none, 1 relu
    int tanh_from;        // channels >= tanh_from get tanh (INT_MAX: none)
    int n_ch_tiles;
    int k_slices;         // > 1: split-K (partial slabs + splitk_reduce_f16)
    int ksteps_per_slice;
    int partial_ld;       // row stride of the partial slabs in floats (coutPad)
    float* partial;       // [k_slices][M][partial_ld] f32 workspace
};

// (ids 4, 9-11, 14, 17, 25, 26 belonged to retired experiments: a 256x128 ring tile, the X3W2 ring, the shared-patch 3x3 kernel, rings of
// four, the fp8 ring of three, register-fed 32 x 32 tiles - DESIGN.md §4, §12; tools/study/retired_r05_forms.patch)
enum ConvTile { TILE_128x128 = 0, TILE_64x256 = 1, TILE_32x256 = 2, TILE_64x256_SMALLC = 3, TILE_128x256 = 5, TILE_256x256 = 6, TILE_128x128_S3 = 7, TILE_256x256_M16 = 8,
                TILE_128x128_M16 = 12, TILE_128x128_S3_M16 = 13, TILE_128x256_M16 = 15, TILE_64x64_S3 = 16, TILE_256x256_FP8 = 20, TILE_128x128_K1 = 21, TILE_64x256_K1 = 22, TILE_128x128_FP8 = 23, TILE_64x64_FP8 = 24,
                TILE_96x128_K1 = 27 /* the streaming tile for a 96-channel remainder (the shared head's channels 256 .. 351): multi-level form only */ };
int conv_tile_ch(ConvTile t);
int conv_tile_m(ConvTile t);
const char* conv_tile_symbol(ConvTile t);
hipError_t launch_conv(const ConvParams& p, ConvTile tile, hipStream_t stream);
hipError_t launch_splitk_reduce(const ConvParams& p, hipStream_t stream);

// A bottleneck block's 3x3 conv, its 1x1 expand conv with the residual add, and the NEXT block's 1x1 reduce conv as one
// kernel (bneck.hip): b stays in LDS, y and a' are written once. planes in {64, 128}.
struct BneckParams {
    const half_t* a;        // [N][H][W][planes]: the block's conv_a output
    unsigned a_bytes, a_zero_off;   // buffer-descriptor range of a (includes the allocation's zero block) and that block's offset
    long long a_img_stride;
    int N, H, W, P, Q, stride, M;   // 3x3, pad 1; M = N * P * Q
    const half_t* w2;       // [planes][9 planes], K index = (r * 3 + s) * planes + c
    unsigned w2_bytes;
    const float* bias2;     // [planes]
    const half_t* w3;       // [4 planes][planes]
    unsigned w3_bytes;
    const float* bias3;     // [4 planes]
    const half_t* res;      // [M][4 planes] dense: the block input (identity shortcut)
    half_t* y;              // [M][4 planes] dense: the block output
    const half_t* w1n;      // [planes][4 planes]: the next block's conv_a (nullptr: none)
    unsigned w1n_bytes;
    const float* bias1n;    // [planes]
    half_t* a_next;         // [M][planes] dense (nullptr: none)
    // the stage's first block: no identity shortcut (res = nullptr); the expand conv's K continues over C2 channels of x2
    // [n][H2][W2][C2] read at stride2, w3 = [4 planes][planes + C2], bias3 = bias_c + bias_d (ConvParams::x2's two-source form)
    const half_t* x2;
    unsigned x2_bytes;
    long long x2_img_stride;
    int W2, C2, stride2;
    // the no-3x3 form (256 planes): a = the block's conv_b OUTPUT [M][planes] (dense rows), w2 / bias2 unused; a' may (also) be
    // written as E4M3 codes (fp8 precision: it feeds an fp8 convolution)
    int no_b;
    unsigned res_bytes;     // ... its residual rows travel by LDS-DMA: the buffer-descriptor range of res
    uint8_t* a_next8;
    const float* a_next8_inv;   // [planes] reciprocal scales, one per channel of a'
};
hipError_t launch_bneck(const BneckParams& p, int planes, int tm, hipStream_t stream);
const char* bneck_symbol(int planes, int tm, bool next, bool dual = false);

// Fused stem: 7x7 stride-2 conv (3 -> 64 channels, bias, ReLU) + 3x3 stride-2 max pool, one kernel.
struct StemPoolParams {
    const half_t* x;      // [n][Hp][Wp][4] f16, image at (+3, +3) inside a zero border (preprocess_rgb8_f16)
    const uint8_t* rgb;   // if not null: the raw frames [n][S][S][3] instead of x; the patch loader normalises them itself
    int S;                //   (same expression as preprocess_rgb8_f16; the border is exact zero)
    const half_t* w;      // stem panel [64][256]: K index = r * 32 + s * 4 + c, zero for s == 7, c == 3, r == 7
    const float* bias;    // [64]
    half_t* pool;         // [n][PO][PO][64]
    half_t* stem;         // optional [n][SO][SO][64] (test hook; nullptr in production runs)
    int n, Hp, Wp, SO, PO, tiles_y, tiles_x;
    long long x_img_stride, pool_img_stride, stem_img_stride;
};
hipError_t launch_stem_pool(const StemPoolParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Element-wise / gather kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------------
hipError_t launch_preprocess(const uint8_t* rgb, half_t* out4, int n, int S, int Hp, int Wp, hipStream_t s);
hipError_t launch_maxpool3x3s2(const half_t* x, half_t* y, int n, int h, int w, int c, int ho, int wo, hipStream_t s);
// y and / or y8 (E4M3 of the f16 result * y8_inv_scale) may be written
hipError_t launch_bilinear(const half_t* x, half_t* y, int n, int h, int w, int c, int ho, int wo,
                           long long x_img_stride, long long y_img_stride, hipStream_t s, uint8_t* y8 = nullptr, const float* y8_inv = nullptr);
// fp8 precision helpers: weight rows -> E4M3 with one scale per row; |x| maximum of an f16 range (as the bit pattern
// of a non-negative float, combined with atomicMax); E4M3 codes -> f32 (debug reads)
// (the input tensor's per-channel activation scales col_scale[C] are folded into the K axis first: K index = tap * C + c)
hipError_t launch_quantize_rows_e4m3(const half_t* x, uint8_t* y, int rows, int K, int C, const float* col_scale, const float* inv_scale_rows, hipStream_t s);
hipError_t launch_rowmax_scaled_f16(const half_t* x, int rows, int K, int C, const float* col_scale, unsigned* out_bits, hipStream_t s);
hipError_t launch_absmax_f16(const half_t* x, long long n, unsigned* out_bits, hipStream_t s);
hipError_t launch_absmax_channels_f16(const half_t* x, long long rows, int C, unsigned* out_bits, hipStream_t s);
hipError_t launch_dequant_e4m3_f32(const uint8_t* x, float* y, long long n, const float* scale_ch, int C, hipStream_t s);
// f32 -> E4M3 code, round to nearest even, saturating at +-448 (NaN keeps its sign with the NaN code)
__device__ __forceinline__ unsigned e4m3_code(float v) {
    if (v != v) return ((__float_as_uint(v) >> 24) & 0x80u) | 0x7Fu;
    const float c = fminf(fmaxf(v, -448.0f), 448.0f);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.0f, 0, false) & 0xFFu;
}
// four codes as one dword (a in the low byte): the same function as four e4m3_code calls, with the packed conversion - two values per
// v_cvt_pk_fp8_f32, clamped by v_med3_f32 - instead of a convert, a mask, a shift and an or per value; NaNs (never seen on a calibrated
// network: the calibration refuses a forward that overflowed) take the element-wise path so that the codes stay identical
__device__ __forceinline__ unsigned e4m3_pack4(float a, float b, float c, float d) {
    if (__builtin_expect(a != a || b != b || c != c || d != d, 0))
        return e4m3_code(a) | (e4m3_code(b) << 8) | (e4m3_code(c) << 16) | (e4m3_code(d) << 24);
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
// heads [n][cells][ldh] f16 -> loc/conf/mask/cells split as f32 (output reads only)
hipError_t launch_split_heads(const half_t* heads, int n, int cells, int ldh, int C, float* loc,
                              float* conf, float* mask, hipStream_t s);
hipError_t launch_f16_to_f32(const half_t* x, float* y, long long n, hipStream_t s);
hipError_t launch_quantize_e4m3(const half_t* x, uint8_t* y, long long n, float inv_scale, hipStream_t s);
// one lane writes 0 to *w: the one-node second branch of captures that have no fork of their own (engine.hip, enqueue_all)
hipError_t launch_side_touch(unsigned* w, hipStream_t s);
hipError_t launch_cells_f32(const half_t* heads, int n, int cells_img, int cells_l0, int ldh, int C,
                            float* out, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Detection tail (detect.hip)
// ---------------------------------------------------------------------------------------------
struct DetectParams {
    const half_t* heads;  // [n][cells][ldh]: 12 box | 3*C conf | 96 mask per cell
    const half_t* proto;  // [n][hp][wp][32]
    const float* priors;  // [P][4]
    int n, P, cells, ldh, C, hp, wp, top_k, max_dets;
    float conf_thresh, nms_thresh;
    int k1_generic;       // 1: the generic softmax/candidate kernel also for 81 classes (yh_tuning.k1_generic)
    // workspaces
    int* cls_count;       // [n][C-1]
    uint2* cand;          // [n][C-1][P]  {score bits, prior}
    float* surv_score;    // [n][(C-1)*top_k]  (-1: empty)
    int* surv_prior;      // same
    float* surv_box;      // [..][4]
    // outputs
    int* det_count;       // [n]
    yh_detection* dets;   // [n][max_dets]
    float* det_crop;      // [n][max_dets][4]  xa, xb, ya, yb
    float* det_coef;      // [n][max_dets][32] the detections' mask coefficients as f32 (det_frame_top -> det_masks)
    uint8_t* masks;       // [n][max_dets][hp*wp]
};
hipError_t launch_detect(const DetectParams& p, hipStream_t s);
int detect_launch_count();
const char* detect_stage_name(int stage);
hipError_t launch_detect_stage(const DetectParams& p, int stage, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Reference-compat kernels (refpath.hip): yolact.rs pre/post-processing on device
// ---------------------------------------------------------------------------------------------
hipError_t launch_resize_v_u32(const uint32_t* src, int sw, int sh, float* tmp, int dh, hipStream_t s);
hipError_t launch_resize_v_rgb8(const uint8_t* src, int sw, int sh, float* tmp, int dh, hipStream_t s);
// horizontal pass; mode 0: RGB8 image [dh][dw][3]; mode 1: tile-split RGB8 [dw/S][S][S][3] with
// S = dh; mode 2: packed u32 r<<24|g<<16|b<<8
hipError_t launch_resize_h(const float* tmp, int sw, int dh, void* dst, int dw, int mode, hipStream_t s);
// postprocess (yolact.rs:90-131) per tile; heads-derived logits; writes cell codes and flags
hipError_t launch_cells_postprocess(const float* cells, int n_tiles, int grid, int C, int mode,
                                    uint32_t* codes, int* diverged, hipStream_t s);
// x8 nearest upsample of cell codes into a [n_tiles][S][S] u32 image, or stitched [S][n_tiles*S]
hipError_t launch_upsample_codes(const uint32_t* codes, int n_tiles, int grid, uint32_t* out,
                                 int stitched, hipStream_t s);

// device spec functions shared by kernels (bit-exact restatement of DESIGN.md §Spec-exp)
__device__ __forceinline__ float spec_expf(float x) {
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    float t = __fmul_rn(x, 1.44269504088896341f);
    float n = rintf(t);
    float r = __fmaf_rn(n, -0.693359375f, x);
    r = __fmaf_rn(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __fmaf_rn(p, r, 1.3981999507e-3f);
    p = __fmaf_rn(p, r, 8.3334519073e-3f);
    p = __fmaf_rn(p, r, 4.1665795894e-2f);
    p = __fmaf_rn(p, r, 1.6666665459e-1f);
    p = __fmaf_rn(p, r, 5.0000001201e-1f);
    float r2 = __fmul_rn(r, r);
    float y = __fmaf_rn(p, r2, r);
    y = __fadd_rn(y, 1.0f);
    int bits = __float_as_int(y) + ((int)n << 23);
    return __int_as_float(bits);
}
// natural logarithm of a positive normal float (scene back-end's sigmoid bump: pow(a, e) = exp(e * log(a))).
// Identical source in oracle/orc_scene.c (it is the spec).
__device__ __forceinline__ float spec_logf(float x) {
    unsigned b = __float_as_uint(x);
    int ex = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = __fmul_rn(m, 0.5f); ex += 1; }
    const float f = __fsub_rn(m, 1.0f), s = __fdiv_rn(f, __fadd_rn(2.0f, f)), z = __fmul_rn(s, s);
    float p = __fmaf_rn(z, 0.11111111f, 0.14285715f);
    p = __fmaf_rn(p, z, 0.2f);
    p = __fmaf_rn(p, z, 0.33333334f);
    const float s2 = __fadd_rn(s, s);
    const float r = __fmaf_rn(__fmul_rn(s2, z), p, s2);
    return __fmaf_rn((float)ex, 0.69314718f, r);
}
__device__ __forceinline__ float spec_tanhf(float x) {
    float a = fabsf(x);
    float e = spec_expf(__fmul_rn(-2.0f, a));
    float t = __fdiv_rn(__fsub_rn(1.0f, e), __fadd_rn(1.0f, e));
    return x < 0.0f ? -t : t;
}

}  // namespace yh
