// tflite_exec.hip — executes a parsed TFLite model (uint8 per-tensor quantised MobileNetV2-style
// graphs: the reference's FRC_model.tflite family, data/README.md:5-16) on the GPU.
//
// SURVEY.md §8f-1. Replaces, for a user who holds the reference's model file, the whole of
// interpreter.invoke() (/root/reference/src/yolact.rs:163) plus the surrounding classify
// (:192-234). The arithmetic restates TensorFlow Lite's published uint8 reference kernels
// (gemmlowp fixed-point requantisation); the runtime itself is the un-vendored tflite 0.9.0 crate
// (Cargo.lock:1106-1108), so parity with it is UNPINNED; tests check this executor bit for bit
// against oracle/tfl_oracle.py on synthetic models.
//
// These are HBM/latency-bound byte kernels on small tensors (a 224x224 MobileNetV2 is ~0.3 GMAC):
// one lane per output element, output channel fastest (coalesced NHWC stores, broadcast input
// reads), int32 accumulation. They are deliberately NOT reshaped into MFMA GEMMs.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "tflite_model.h"
#include "yh_internal.h"

using namespace yh;

namespace {

__device__ __forceinline__ int q_srdhm(int a, int b) {
    if (a == INT_MIN && b == INT_MIN) return INT_MAX;
    const long long ab = (long long)a * (long long)b;
    const long long nudge = ab >= 0 ? (1ll << 30) : (1ll - (1ll << 30));
    return (int)((ab + nudge) / (1ll << 31));
}
__device__ __forceinline__ int q_rdbpot(int x, int e) {
    if (e == 0) return x;
    const int mask = (1 << e) - 1, rem = x & mask, thr = (mask >> 1) + (x < 0 ? 1 : 0);
    return (x >> e) + (rem > thr ? 1 : 0);
}
__device__ __forceinline__ int q_mbqm(int x, int m, int shift) {
    const int left = shift > 0 ? shift : 0, right = shift > 0 ? 0 : -shift;
    return q_rdbpot(q_srdhm(x * (1 << left), m), right);
}
__device__ __forceinline__ int q_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Operators folded into their producer at plan time (round 4, yh_tuning.tfl_fuse): the producer computes its own uint8 output
// value exactly as TFLite does - in a register - and the element-wise operators that consumed it (QUANTIZE / RELU / RELU6 as a
// requantisation, TANH as its 256-entry table, ADD with its other operand read from memory) run on that value before the one
// store: the same integers in the same order, no intermediate tensor, no launch. A CONCATENATION part with the output's own
// quantisation is written by its producer straight into the concatenated tensor, a PAD in front of a convolution becomes that
// convolution's padding (a padded tap holds the zero point: (x - zx) = 0, the tap TFLite's kernels skip).
struct PostStep { int kind; int zi, zo, m, s, lo, hi; const uint8_t* lut; };   // 1 requantise, 2 table, 3 the ADD below
struct PostOps {
    int n;
    PostStep st[3];
    const uint8_t* other; long long other_s;    // ADD: the other operand (same shape), bytes per image
    int q_is_a, za, zb, m1, s1, m2, s2, mo, so, azo, alo, ahi;
};
__device__ __forceinline__ int apply_post(const PostOps& po, int q, long long img, long long elem) {
    // (compile-time step indices: a run-time index into the kernel-argument block makes the compiler copy the whole block to
    // scratch - 344 bytes of private memory per lane in every convolution kernel, +45 % per invoke when first measured)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i >= po.n) break;
        const PostStep& t = po.st[i];
        if (t.kind == 1) q = q_clamp(q_mbqm(q - t.zi, t.m, t.s) + t.zo, t.lo, t.hi);
        else if (t.kind == 2) q = t.lut[q];
        else {
            const int o = po.other[img * po.other_s + elem];
            const int a = po.q_is_a ? q : o, b = po.q_is_a ? o : q;
            const int v1 = q_mbqm((a - po.za) * (1 << 20), po.m1, po.s1), v2 = q_mbqm((b - po.zb) * (1 << 20), po.m2, po.s2);
            q = q_clamp(q_mbqm(v1 + v2, po.mo, po.so) + po.azo, po.alo, po.ahi);
        }
    }
    return q;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v4i_u __attribute__((ext_vector_type(4), aligned(4)));   // (a 16-byte load from a dword-aligned address)
struct ConvQ {
    const uint8_t *x, *w; const int* bias; uint8_t* y;
    int H, W, Ci, Ho, Wo, Co, kh, kw, sh, sw, ph, pw, dh, dw, dm;
    int zx, zw, zo, mult, shift, lo, hi;
    const int* wsum;   // [Co][kh*kw]: sum of the raw weight bytes of one tap (dot-product kernel), or nullptr
    // batch plan (yh_tfl_set_batch): the grid's last used dimension is the image; activations are image-major
    long long xs, ys;  // bytes per image of x / y
    PostOps po;
};

__global__ __launch_bounds__(256) void tfl_conv_u8(const ConvQ p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.Ho * p.Wo * p.Co) return;
    const uint8_t* const px = p.x + blockIdx.y * p.xs;
    uint8_t* const py = p.y + blockIdx.y * p.ys;
    const int oc = t % p.Co, r0 = t / p.Co, ox = r0 % p.Wo, oy = r0 / p.Wo;
    int acc = 0;
    for (int r = 0; r < p.kh; ++r) {
        const int iy = oy * p.sh - p.ph + r * p.dh;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int s = 0; s < p.kw; ++s) {
            const int ix = ox * p.sw - p.pw + s * p.dw;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            const uint8_t* xp = px + ((size_t)iy * p.W + ix) * p.Ci;
            const uint8_t* wp = p.w + (((size_t)oc * p.kh + r) * p.kw + s) * p.Ci;
            for (int c = 0; c < p.Ci; ++c) acc += ((int)xp[c] - p.zx) * ((int)wp[c] - p.zw);
        }
    }
    acc += p.bias ? p.bias[oc] : 0;
    py[t] = (uint8_t)apply_post(p.po, q_clamp(q_mbqm(acc, p.mult, p.shift) + p.zo, p.lo, p.hi), blockIdx.y, t);
}

// The same for layers the dot-product and MFMA kernels cannot take (Ci % 4 != 0: the model's first convolution, 3 channels) when
// yh_tuning.tfl_dot >= 1: one lane = one pixel x 8 output channels instead of one output element. The workgroup (256 pixels x 8
// channels) first puts its 8 x K weights, zero point already subtracted, into LDS as [k][8] ints: a lane then loads each input byte
// once for eight MACs and reads its weights with two broadcast ds_read_b128 (the element-per-lane kernel loads an input byte and a
// weight byte per MAC: 19.2 us for the 224 x 224 x 3 -> 112 x 112 x 32 layer of two images; this form: see DESIGN.md section 8).
constexpr int kPx8MaxK = 512;   // taps x input channels a launch of this kernel can hold (16 KB of LDS)
template <int KH, int KW, int CI>   // (0, 0, 0: run-time extent; 3, 3, 3: the RGB stem)
__global__ __launch_bounds__(256) void tfl_conv_u8_px8(const ConvQ p) {
    __shared__ __attribute__((aligned(16))) int wl[kPx8MaxK * 8];
    const int K = p.kh * p.kw * p.Ci, oc0 = blockIdx.y * 8;
    for (int i = threadIdx.x; i < K * 8; i += 256) {
        const int k = i >> 3, j = i & 7, oc = oc0 + j < p.Co ? oc0 + j : p.Co - 1;   // (the clamped duplicates are not stored)
        wl[i] = (int)p.w[(size_t)oc * K + k] - p.zw;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.Ho * p.Wo) return;
    const uint8_t* const px = p.x + blockIdx.z * p.xs;
    uint8_t* const py = p.y + blockIdx.z * p.ys;
    const int ox = pix % p.Wo, oy = pix / p.Wo;
    int acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if constexpr (KH > 0) {   // extent known at compile time: every input byte of the lane is asked for before the first MAC
        int xv[KH * KW * CI];
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const int iy = oy * p.sh - p.ph + r * p.dh, ix = ox * p.sw - p.pw + s * p.dw;
                const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint8_t* xp = px + ((size_t)(in ? iy : 0) * p.W + (in ? ix : 0)) * CI;
#pragma unroll
                for (int c = 0; c < CI; ++c) { const int v = (int)xp[c]; xv[(r * KW + s) * CI + c] = in ? v - p.zx : 0; }
            }
#pragma unroll
        for (int k = 0; k < KH * KW * CI; ++k) {
            const v4i w0 = *(const v4i*)(wl + k * 8), w1 = *(const v4i*)(wl + k * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += xv[k] * w0[j]; acc[4 + j] += xv[k] * w1[j]; }
        }
    } else {
        for (int r = 0; r < p.kh; ++r) {
            const int iy = oy * p.sh - p.ph + r * p.dh;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int s = 0; s < p.kw; ++s) {
                const int ix = ox * p.sw - p.pw + s * p.dw;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const uint8_t* xp = px + ((size_t)iy * p.W + ix) * p.Ci;
                const int* wk = wl + (r * p.kw + s) * p.Ci * 8;
                for (int c = 0; c < p.Ci; ++c) {
                    const int xv = (int)xp[c] - p.zx;
                    const v4i w0 = *(const v4i*)(wk + c * 8), w1 = *(const v4i*)(wk + c * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[j] += xv * w0[j]; acc[4 + j] += xv * w1[j]; }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int oc = oc0 + j;
        if (oc < p.Co) {
            const int a = acc[j] + (p.bias ? p.bias[oc] : 0);
            py[(size_t)pix * p.Co + oc] = (uint8_t)apply_post(p.po, q_clamp(q_mbqm(a, p.mult, p.shift) + p.zo, p.lo, p.hi), blockIdx.z, (long long)pix * p.Co + oc);
        }
    }
}

// Same arithmetic, four MACs per instruction: for Ci % 4 == 0 the sum over the valid taps of
// (x - zx)(w - zw) is  sum(x w) - zw sum(x) - zx sum(w) + n zx zw  with the first two as
// v_dot4_u32_u8 over raw bytes and sum(w) per (channel, tap) tabulated at load time — all exact in
// int32 (|sum(x w)| <= 65025 * 9 * 1024 < 2^31 is checked when the plan is built). One lane = one
// pixel x 8 output channels; the channel block is the grid's y index, so weight words are
// wave-uniform and arrive through the scalar cache.
// A workgroup = 64 pixels x 8 channels; with KS == 4 its four waves each take a quarter of the input
// channels (the model's layers are small: 28 x 28 pixels x 128 channels is 13 x 16 workgroups, and
// the serial chain per lane is what takes the time) and wave 0 adds the partial sums from LDS.
template <int KS>
__global__ __launch_bounds__(64 * KS) void tfl_conv_u8_dot(const ConvQ p) {
    __shared__ unsigned part[KS > 1 ? (KS - 1) * 9 * 64 : 1];
    const uint8_t* const px = p.x + blockIdx.z * p.xs;
    uint8_t* const py = p.y + blockIdx.z * p.ys;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int pix = blockIdx.x * 64 + lane, oc0 = blockIdx.y * 8;
    const bool live = pix < p.Ho * p.Wo;
    const int ox = live ? pix % p.Wo : 0, oy = live ? pix / p.Wo : 0, ci4 = p.Ci >> 2, ntaps = p.kh * p.kw;
    const int c_lo = ks * (ci4 / KS), c_hi = c_lo + ci4 / KS;
    const unsigned* w32 = (const unsigned*)p.w;
    unsigned acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, sx = 0;
    int ws[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, nv = 0;
    for (int r = 0; r < p.kh; ++r) {
        const int iy = oy * p.sh - p.ph + r * p.dh;
        for (int s = 0; s < p.kw; ++s) {
            const int ix = ox * p.sw - p.pw + s * p.dw, tap = r * p.kw + s;
            if (!live || (unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
            const unsigned* xp = (const unsigned*)(px + ((size_t)iy * p.W + ix) * p.Ci);
            for (int c = c_lo; c < c_hi; ++c) {
                const unsigned xv = xp[c];
                sx = __builtin_amdgcn_udot4(xv, 0x01010101u, sx, false);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int oc = oc0 + j < p.Co ? oc0 + j : p.Co - 1;   // wave-uniform; the clamped duplicates are not stored
                    acc[j] = __builtin_amdgcn_udot4(xv, w32[((size_t)oc * ntaps + tap) * ci4 + c], acc[j], false);
                }
            }
            if (ks == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ws[j] += p.wsum[(oc0 + j < p.Co ? oc0 + j : p.Co - 1) * ntaps + tap];
                ++nv;
            }
        }
    }
    if (KS > 1) {
        if (ks > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part[((ks - 1) * 9 + j) * 64 + lane] = acc[j];
            part[((ks - 1) * 9 + 8) * 64 + lane] = sx;
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += part[(k * 9 + j) * 64 + lane];
            sx += part[(k * 9 + 8) * 64 + lane];
        }
    }
    if (!live) return;
    const int base = nv * p.Ci * p.zx * p.zw - p.zw * (int)sx;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int oc = oc0 + j;
        if (oc < p.Co) {
            const int a = (int)acc[j] + base - p.zx * ws[j] + (p.bias ? p.bias[oc] : 0);
            py[(size_t)pix * p.Co + oc] = (uint8_t)apply_post(p.po, q_clamp(q_mbqm(a, p.mult, p.shift) + p.zo, p.lo, p.hi), blockIdx.z, (long long)pix * p.Co + oc);
        }
    }
}

// ---- CONV_2D on the int8 matrix pipes: Ci % 64 == 0 (the FPN / protonet / head 3x3 convolutions of the
// model family - 85 % of its MACs - and the wide 1x1 convolutions). uint8 operands become int8 by flipping the top bit
// (x' = x - 128, w' = w - 128), v_mfma_i32_16x16x64_i8 accumulates sum(x' w') exactly in int32, and
//   sum (x - zx)(w - zw) = sum x'w' + (128 - zw) sum x' + (128 - zx) sum w' + K (128 - zx)(128 - zw)
// restores TFLite's value bit for bit: sum w' per channel is tabulated at load time, sum x' per output pixel is a v_dot4 beside
// the MFMAs, and a padded tap is fed the input's zero point (x - zx = 0: TFLite skips it). Implicit GEMM: a workgroup owns
// 64 channels x 64 output pixels (images of the batch plan are folded into the pixel index), four waves of 32 x 32, one tap's
// 64 channels per k-step, register-staged double buffer (tiles are 4 KB: the layers are small and latency-bound).
struct ConvI8 {
    const uint8_t* x; const uint8_t* wq; const int* cterm; uint8_t* y;   // wq: [CoPad][K] bytes w ^ 0x80, K = (r, s, c); cterm: [CoPad]
    int H, W, Ci, Ho, Wo, Co, kh, kw, sh, sw, ph, pw, dh, dw;
    int zx, zw, zo, mult, shift, lo, hi, K, M;
    long long xs, ys;
    PostOps po;
};
// Epilogue of the int8 MFMA convolutions: lane = pixel (wm, j, l15), channels (wc, i, 4 lg + e) [C/D layout of the 16 x 16 MFMA:
// column = lane & 15, rows 4 (lane >> 4) + e]; sx[j] = sum of the raw input bytes over the whole K of pixel (wm, j, l15).
template <int TI, int TJ>   // MFMA tiles per wave along the channels / the pixels
__device__ __forceinline__ void conv_i8_epilogue(const ConvI8& p, const v4i (&acc)[TI][TJ], const unsigned (&sx)[TJ], int m0, int ch0, int wc, int wm, int l15, int lg, int HoWo) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int mo = m0 + wm * 16 * TJ + j * 16 + l15;
        if (mo >= p.M) continue;
        const int im = mo / HoWo, px = mo - im * HoWo;
        uint8_t* yrow = p.y + (long long)im * p.ys + (size_t)px * p.Co;
        const int xterm = (128 - p.zw) * ((int)sx[j] - 128 * p.K);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ch = ch0 + wc * 16 * TI + i * 16 + 4 * lg;
            if (ch >= p.Co) continue;
            unsigned packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = acc[i][j][e] + xterm + p.cterm[ch + e];   // (cterm is padded to CoPad)
                const unsigned q = (unsigned)apply_post(p.po, q_clamp(q_mbqm(a, p.mult, p.shift) + p.zo, p.lo, p.hi), im, (long long)px * p.Co + (ch + e < p.Co ? ch + e : 0));
                packed |= q << (8 * e);
            }
            if ((p.Co & 3) == 0 && (((size_t)yrow) & 3) == 0) *(unsigned*)(yrow + ch) = packed;   // (a CONCATENATION part may start at any byte)
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (ch + e < p.Co) yrow[ch + e] = (uint8_t)(packed >> (8 * e));
            }
        }
    }
}


__global__ __launch_bounds__(256) void tfl_conv_i8_mfma(const ConvI8 p) {
    __shared__ __attribute__((aligned(16))) char lds[2][8192];   // [stage][A 64 rows x 64 B | B 64 rows x 64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.x * 64, ch0 = blockIdx.y * 64;
    // loader role: 16-byte chunk `lchunk` of row `lrow` of both tiles
    const int lrow = tid >> 2, lchunk = tid & 3;
    const int HoWo = p.Ho * p.Wo;
    const int m = m0 + lrow;
    const bool mlive = m < p.M;
    const int img = mlive ? m / HoWo : 0, rem = mlive ? m - img * HoWo : 0, oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const uint8_t* ximg = p.x + (long long)img * p.xs;
    const uint8_t* wrow = p.wq + (size_t)(ch0 + lrow) * p.K + lchunk * 16;
    const unsigned zx4 = (unsigned)p.zx * 0x01010101u;
    const int lds_w = lrow * 64 + ((lchunk ^ ((lrow >> 2) & 3)) << 4);
    const int cchunks = p.Ci >> 6, nsteps = p.kh * p.kw * cchunks;
    // Register ring: the tiles of the next THREE k-steps are in flight while one is computed. The layers are small (50-200
    // workgroups of 2-18 k-steps) and every step used to wait one L2 round trip for the single tile it had prefetched: the kernel
    // was bound by steps x latency (12 us per launch, 46 launches = 0.67 of a 0.97 ms invoke). The ring is rotated by register
    // moves (a move waits for the load that fills its source, so four slots give three steps of distance).
    struct Tile { uint4 a, b; };
    // position of the NEXT tile to fetch along K, advanced incrementally (k order: tap outer, 64-channel chunk inner): the step
    // index is wave-uniform, and two integer divisions per k-step were a third of a step's instructions on a wave that has its
    // SIMD to itself (one workgroup of four waves per CU: nothing else hides them)
    int f_r = 0, f_s = 0, f_cc = 0, f_step = 0;
    const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
    const uint8_t* const xlane = ximg + lchunk * 16;
    auto fetch = [&]() {
        Tile t;
        t.a = *(const uint4*)(wrow + (size_t)f_step * 64);       // K index = tap * Ci + cc * 64 = step * 64
        const int iy = iy0 + f_r * p.dh, ix = ix0 + f_s * p.dw;
        const bool in = mlive && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        // padded tap (and rows past M): x = zx contributes (x - zx) = 0. (The address is clamped into the image, the value selected.)
        const uint4 v = *(const uint4*)(xlane + (unsigned)(((in ? iy : 0) * p.W + (in ? ix : 0)) * p.Ci + f_cc * 64));
        t.b = in ? v : make_uint4(zx4, zx4, zx4, zx4);
        if (f_step + 1 < nsteps) {                                // (past the end: the last tile again, never used)
            ++f_step;
            if (++f_cc == cchunks) { f_cc = 0; if (++f_s == p.kw) { f_s = 0; ++f_r; } }
        }
        return t;
    };
    auto stash = [&](int st, const Tile& t) {
        *(uint4*)(lds[st] + lds_w) = t.a;
        *(uint4*)(lds[st] + 4096 + lds_w) = t.b;
    };
    const int wc = wave >> 1, wm = wave & 1;
    v4i acc[2][2];
    unsigned sx[2] = { 0u, 0u };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4i{ 0, 0, 0, 0 };
    Tile t0 = fetch(), t1 = fetch(), t2 = fetch(), t3 = fetch();
    stash(0, t0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int st = step & 1;
        // t1, t2, t3 = tiles step + 1 .. step + 3 (in flight); the slot of the tile now in LDS takes tile step + 4
        t0 = fetch();
        v4i fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wc * 32 + i * 16 + l15;
            fa[i] = *(const v4i*)(lds[st] + row * 64 + ((lg ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wm * 32 + j * 16 + l15;
            const v4i raw = *(const v4i*)(lds[st] + 4096 + row * 64 + ((lg ^ ((row >> 2) & 3)) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sx[j] = __builtin_amdgcn_udot4((unsigned)raw[e], 0x01010101u, sx[j], false);
                fb[j][e] = raw[e] ^ (int)0x80808080u;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
        stash(st ^ 1, t1);                                     // (the other stage was last read one step ago, behind a barrier; after the last step: unused)
        __syncthreads();
        t1 = t2; t2 = t3; t3 = t0;
    }
    // sum x over the whole K of this lane's pixel: the four lane groups hold the four 16-byte quarters of every k-step
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sx[j] += __shfl_xor(sx[j], 16);
        sx[j] += __shfl_xor(sx[j], 32);
    }
    conv_i8_epilogue<2, 2>(p, acc, sx, m0, ch0, wc, wm, l15, lg, HoWo);
}

// ---- The same convolution with NO LDS and no barrier (yh_tuning.tfl_dot = 3, the default, for the launches conv_i8_direct_pays()
// names): one wave = ONE 16 x 16 MFMA tile, and every operand register is one 16-byte global load - lane (l15, lg) of
// v_mfma_i32_16x16x64_i8 holds bytes 16 lg .. 16 lg + 15 of row l15's 64-deep K chunk, which is contiguous both in the weight panel
// [Co][K] and in an NHWC pixel's channels. What a launch of the LDS tiles costs (tools/study/tfl_layer_table.py: the per-launch
// timeline of the 136-op model) is not its MFMAs: it is the serial k-loop (0.36 us per step: stash / barrier / fragment reads on
// waves that have their SIMD to themselves) and the epilogue - 16 outputs per lane, each a fixed-point requantisation and, where an
// ADD or a LUT is folded in, that arithmetic too (a projection convolution with the residual ADD folded in: 15.6 us against 8.5
// without) - on 2 to 50 workgroups of a 256-CU chip. Here a lane finishes 4 outputs, a 7-workgroup layer becomes 100 waves, a k-step
// is 2 loads, 4 v_dot4, 4 v_xor and 1 MFMA, and D steps of loads are in flight (a ring of registers indexed at compile time). The
// re-reads of the operands (each weight row by every pixel tile, each pixel by every channel tile) are L2 hits. K chunks past Ci
// (Ci % 4 == 0, not % 64: 16, 24, 32, 96, 144 channels) are fed zeros on both sides, which add nothing to sum x'w' nor to sum x.
// D = the depth of the register ring = the k-steps of one loop iteration. The loop body has NO branch: with a wave-uniform
// `if (step < nsteps)` around each step hipcc's wait insertion gave up counting and put s_waitcnt vmcnt(0..3) in front of every step
// - eight steps of loads "in flight" that were waited for one by one. So the k-steps are rounded up to a multiple of D (the launch
// picks the D that pads least), and the steps past the end load the last step again and are fed zeros.
// ONCE: the launch has exactly D k-steps - every load of the wave is asked for before its first MFMA and nothing is fetched again
// (the 3x3 x 128-channel head convolutions, 18 steps: 9.5 -> 7 us against two drained iterations of a ring of 9).
// KK (ONCE only): the kernel extent, 1 x 1 or 3 x 3 - the tap and the channel chunk of step d are then compile-time constants and
// the position bookkeeping below disappears from the prologue (0: run-time extent, looped form).
template <int D, bool ONCE, int KK>
__device__ __forceinline__ void conv_i8_direct_tile(const ConvI8& p, int tile_m, int tile_c) {
    static_assert(KK == 0 || (ONCE && D % (KK * KK) == 0), "compile-time extents: the launch's k-steps are exactly D");
    const int lane = threadIdx.x, l15 = lane & 15, lg = lane >> 4;
    const int m0 = tile_m * 16, ch0 = tile_c * 16;
    const int HoWo = p.Ho * p.Wo;
    const int m = m0 + l15;
    const bool live = m < p.M;
    const int img = live ? m / HoWo : 0, rem = live ? m - img * HoWo : 0, oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const uint8_t* const xb = p.x + (long long)img * p.xs;
    const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
    const uint8_t* const wa = p.wq + (size_t)(ch0 + l15) * p.K;   // (rows up to CoPad exist)
    const int cchunks = (p.Ci + 63) >> 6, nsteps = p.kh * p.kw * cchunks;
    const int zx4 = (int)((unsigned)p.zx * 0x01010101u);
    struct Tile { v4i a, b; };
    int f_r = 0, f_s = 0, f_cc = 0, f_tap = 0, f_n = 0;   // the next k-step to fetch: tap (f_r, f_s) = f_tap, 64-channel chunk f_cc
    auto fetch_at = [&](int r, int s_, int tap, int cc, bool more) {
        Tile t;
        // (per lane group: the last chunk of a Ci that is not a multiple of 64 holds nd < 4 dwords of this lane's 16 bytes - the rest
        // belongs to the next tap / pixel and is fed zeros; Ci % 4 == 0, so the loads are dword-aligned, not 16-byte aligned)
        const int left = p.Ci - (cc * 64 + lg * 16), nd = !more ? 0 : (left >= 16 ? 4 : (left > 0 ? left >> 2 : 0));
        const int coff = left > 0 ? cc * 64 + lg * 16 : 0;   // (a lane group past Ci reads its row's first bytes: nothing is read more than 12 bytes past a row)
        const v4i va = *(const v4i_u*)(wa + tap * p.Ci + coff);
        const int iy = iy0 + r * p.dh, ix = ix0 + s_ * p.dw;
        const bool in = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        // padded tap (and rows past M): x = zx contributes (x - zx) = 0. (The address is clamped into the image, the value selected.)
        const v4i vb = *(const v4i_u*)(xb + (unsigned)(((in ? iy : 0) * p.W + (in ? ix : 0)) * p.Ci + coff));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t.a[e] = e < nd ? va[e] : 0;
            t.b[e] = e < nd ? (in ? vb[e] : zx4) : 0;
        }
        return t;
    };
    auto fetch = [&](int d) {
        if constexpr (KK > 0) {
            constexpr int CC = D / (KK * KK);
            const int tap = d / CC;
            return fetch_at(tap / KK, tap % KK, tap, d % CC, true);
        } else {
            // (wave-uniform position; past the end it stays on the last step and the tile is zeros)
            const Tile t = fetch_at(f_r, f_s, f_tap, f_cc, f_n < nsteps);
            // advance (selects, no branch); the last step is never left
            const int adv = f_n + 1 < nsteps;
            f_n += f_n < nsteps;
            const int ncc = f_cc + 1, wrap_c = ncc == cchunks;
            const int ns = f_s + 1, wrap_s = wrap_c && ns == p.kw;
            f_cc = adv ? (wrap_c ? 0 : ncc) : f_cc;
            f_tap = adv ? f_tap + wrap_c : f_tap;
            f_s = adv ? (wrap_c ? (wrap_s ? 0 : ns) : f_s) : f_s;
            f_r = adv ? f_r + wrap_s : f_r;
            return t;
        }
    };
    v4i acc[1][1] = { { v4i{ 0, 0, 0, 0 } } };
    unsigned sx[1] = { 0u };
    Tile ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = fetch(d);
    for (int step = 0; step < (ONCE ? 1 : nsteps); step += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const Tile t = ring[d];
            if (!ONCE) ring[d] = fetch(d);   // (k-step step + d + D: asked for before this step's MFMA is issued)
            v4i fb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sx[0] = __builtin_amdgcn_udot4((unsigned)t.b[e], 0x01010101u, sx[0], false);
                fb[e] = t.b[e] ^ (int)0x80808080u;
            }
            acc[0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(t.a, fb, acc[0][0], 0, 0, 0);
        }
    }
    sx[0] += __shfl_xor(sx[0], 16);
    sx[0] += __shfl_xor(sx[0], 32);
    conv_i8_epilogue<1, 1>(p, acc, sx, m0, ch0, 0, 0, l15, lg, HoWo);
}
template <int D, bool ONCE, int KK>
__global__ __launch_bounds__(64) void tfl_conv_i8_direct(const ConvI8 p) {
    conv_i8_direct_tile<D, ONCE, KK>(p, blockIdx.x, blockIdx.y);
}
// Several INDEPENDENT convolutions of one kernel form as ONE launch (yh_tuning.tfl_group; group_plan): the prediction head's
// tower convolutions of the five pyramid levels, then its fifteen output convolutions, the FPN's output convolutions. The
// problems' parameter blocks sit in device memory; a workgroup finds its problem in the tile prefix table (at most kMaxGroup
// entries, wave-uniform) and runs the same tile code on it.
constexpr int kMaxGroup = 16;
template <int D, bool ONCE, int KK>
__global__ __launch_bounds__(64) void tfl_conv_i8_direct_group(const ConvI8* __restrict__ probs, const int* __restrict__ tile_start, int nprob) {
    const int bid = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < nprob; ++i) pi = bid >= tile_start[i] ? i : pi;
    const ConvI8 p = probs[pi];
    const int t = bid - tile_start[pi], mtiles = (p.M + 15) >> 4;
    conv_i8_direct_tile<D, ONCE, KK>(p, t % mtiles, t / mtiles);
}
// The ring depth for a launch of n k-steps: n itself where a kernel of that depth exists (ONCE), else the D of {9 8 6 5 4} that
// pads n least, the deepest among equals.
static int conv_i8_direct_depth(int n, bool* once) {
    static const int exact[] = { 1, 2, 3, 4, 5, 6, 8, 9, 12, 15, 18 };
    for (int d : exact) if (d == n) { *once = true; return d; }
    *once = false;
    static const int ds[] = { 9, 8, 6, 5, 4 };
    int best = 4, pad = 1 << 30;
    for (int d : ds) { if (d > n) continue; const int q = (n + d - 1) / d * d - n; if (q < pad) { pad = q; best = d; } }
    return best;
}
// One launch of the register-fed kernel: a single convolution (probs == nullptr) or a group of them (device arrays, `tiles` workgroups)
struct DirectLaunch { const ConvI8* q; const ConvI8* probs; const int* tile_start; int nprob; int tiles; };
template <int D, bool ONCE, int KK>
static void launch_conv_i8_direct_d(const DirectLaunch& L, hipStream_t s) {
    if (L.probs) hipLaunchKernelGGL((tfl_conv_i8_direct_group<D, ONCE, KK>), dim3((unsigned)L.tiles), dim3(64), 0, s, L.probs, L.tile_start, L.nprob);
    else hipLaunchKernelGGL((tfl_conv_i8_direct<D, ONCE, KK>), dim3((unsigned)((L.q->M + 15) / 16), (unsigned)((L.q->Co + 15) / 16)), dim3(64), 0, s, *L.q);
}
// The kernel form of a convolution: (D, ONCE, KK) packed into one int (convolutions of one group share it)
static int conv_i8_direct_form(const ConvI8& q) {
    bool once = false;
    const int n = q.kh * q.kw * ((q.Ci + 63) / 64);
    int d = conv_i8_direct_depth(n, &once);
    const int kk = q.kh == 1 && q.kw == 1 ? 1 : (q.kh == 3 && q.kw == 3 ? 3 : 0);
    if (once && kk == 3 && (d == 9 || d == 18)) return d * 16 + 8 + 3;
    if (once && kk == 1) return d * 16 + 8 + 1;
    if (once) d = n >= 9 ? 9 : (n >= 8 ? 8 : (n >= 6 ? 6 : (n >= 5 ? 5 : 4)));   // any other extent (or step count): the looped form
    return d * 16;
}
static void launch_conv_i8_direct(const DirectLaunch& L, hipStream_t s) {
    switch (conv_i8_direct_form(*L.q)) {
        case 9 * 16 + 8 + 3: launch_conv_i8_direct_d<9, true, 3>(L, s); break;
        case 18 * 16 + 8 + 3: launch_conv_i8_direct_d<18, true, 3>(L, s); break;
        case 1 * 16 + 8 + 1: launch_conv_i8_direct_d<1, true, 1>(L, s); break;
        case 2 * 16 + 8 + 1: launch_conv_i8_direct_d<2, true, 1>(L, s); break;
        case 3 * 16 + 8 + 1: launch_conv_i8_direct_d<3, true, 1>(L, s); break;
        case 4 * 16 + 8 + 1: launch_conv_i8_direct_d<4, true, 1>(L, s); break;
        case 5 * 16 + 8 + 1: launch_conv_i8_direct_d<5, true, 1>(L, s); break;
        case 6 * 16 + 8 + 1: launch_conv_i8_direct_d<6, true, 1>(L, s); break;
        case 8 * 16 + 8 + 1: launch_conv_i8_direct_d<8, true, 1>(L, s); break;
        case 9 * 16 + 8 + 1: launch_conv_i8_direct_d<9, true, 1>(L, s); break;
        case 12 * 16 + 8 + 1: launch_conv_i8_direct_d<12, true, 1>(L, s); break;
        case 15 * 16 + 8 + 1: launch_conv_i8_direct_d<15, true, 1>(L, s); break;
        case 18 * 16 + 8 + 1: launch_conv_i8_direct_d<18, true, 1>(L, s); break;
        case 4 * 16: launch_conv_i8_direct_d<4, false, 0>(L, s); break;
        case 5 * 16: launch_conv_i8_direct_d<5, false, 0>(L, s); break;
        case 6 * 16: launch_conv_i8_direct_d<6, false, 0>(L, s); break;
        case 8 * 16: launch_conv_i8_direct_d<8, false, 0>(L, s); break;
        default: launch_conv_i8_direct_d<9, false, 0>(L, s); break;
    }
}
// Which int8 MFMA launches take the register-fed kernel: every one whose input channels the LDS tiles cannot take (Ci % 64 != 0), and
// the small ones - up to 2048 waves (8 per CU), or up to 8 k-steps. A large layer with a long K (the protonet's 3x3 x 128 channels
// at 56 x 56 x 2 images: 3136 waves of 18 steps) re-reads its operands once per 16 x 16 tile and measured 26.0 us against 14.2 on
// the 64 x 64 LDS tiles.
static bool conv_i8_direct_pays(const ConvI8& q) {
    if (q.Ci % 64 != 0) return true;
    const long long waves = (long long)((q.M + 15) / 16) * ((q.Co + 15) / 16);
    return waves <= 2048 || q.kh * q.kw * (q.Ci / 64) <= 8;
}

// (Round 4 also built the same convolution with its K split over the four waves of the workgroup - every wave the whole 64 x 64
// tile for a quarter of the k-steps, partial tiles added through LDS atomics: bit-identical, and SLOWER everywhere (18 k-steps: 13 ->
// 16-20 us, 4 k-steps: 5.3 -> 9.5 us). A launch costs 6.4 us + 0.36 us per k-step whatever its pixel count (tools/study/
// tfl_conv_steps.py), and that per-step cost is the wave's own instruction issue - 88 instructions - not a latency the split
// could overlap: four times the work per wave and step took four times as long. Removed.)

__global__ __launch_bounds__(256) void tfl_dwconv_u8(const ConvQ p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.Ho * p.Wo * p.Co) return;
    const uint8_t* const px = p.x + blockIdx.y * p.xs;
    uint8_t* const py = p.y + blockIdx.y * p.ys;
    const int oc = t % p.Co, r0 = t / p.Co, ox = r0 % p.Wo, oy = r0 / p.Wo, ic = oc / p.dm;
    int acc = 0;
    for (int r = 0; r < p.kh; ++r) {
        const int iy = oy * p.sh - p.ph + r * p.dh;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int s = 0; s < p.kw; ++s) {
            const int ix = ox * p.sw - p.pw + s * p.dw;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            acc += ((int)px[((size_t)iy * p.W + ix) * p.Ci + ic] - p.zx) * ((int)p.w[((size_t)r * p.kw + s) * p.Co + oc] - p.zw);
        }
    }
    acc += p.bias ? p.bias[oc] : 0;
    py[t] = (uint8_t)apply_post(p.po, q_clamp(q_mbqm(acc, p.mult, p.shift) + p.zo, p.lo, p.hi), blockIdx.y, t);
}

// Depth multiplier 1, channels a multiple of 4, dword-aligned tensors, kernel extent known at compile time (3 x 3: every depthwise
// layer of the model family): one lane = one pixel x FOUR channels - a tap is one dword of input and one dword of weights for four
// MACs - and the loop over the taps has no branch: all KH x KW input dwords are asked for at once (a tap outside the image loads
// from a clamped address and is given the zero point, (x - zx) = 0), so a lane waits one memory latency instead of one per tap. The
// element-per-lane kernel above takes 5.2 us (small layers) to 10.7 us (112 x 112 x 32 x 2 images) per launch.
template <int KH, int KW>
__global__ __launch_bounds__(256) void tfl_dwconv_u8_c4(const ConvQ p) {
    const int c4n = p.Co >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.Ho * p.Wo * c4n) return;
    const uint8_t* const px = p.x + blockIdx.y * p.xs + 4 * (t % c4n);
    uint8_t* const py = p.y + blockIdx.y * p.ys;
    const int cg = t % c4n, r0 = t / c4n, ox = r0 % p.Wo, oy = r0 / p.Wo;
    const unsigned zx4 = (unsigned)p.zx * 0x01010101u;
    unsigned xv[KH * KW], wv[KH * KW];
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
        for (int s = 0; s < KW; ++s) {
            const int iy = oy * p.sh - p.ph + r * p.dh, ix = ox * p.sw - p.pw + s * p.dw;
            const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned v = *(const unsigned*)(px + ((size_t)(in ? iy : 0) * p.W + (in ? ix : 0)) * p.Ci);
            xv[r * KW + s] = in ? v : zx4;
            wv[r * KW + s] = *(const unsigned*)(p.w + (size_t)(r * KW + s) * p.Co + 4 * cg);
        }
    int acc[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < KH * KW; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += ((int)((xv[k] >> (8 * e)) & 255u) - p.zx) * ((int)((wv[k] >> (8 * e)) & 255u) - p.zw);
    const long long e0 = (long long)r0 * p.Co + 4 * cg;
    unsigned packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int a = acc[e] + (p.bias ? p.bias[4 * cg + e] : 0);
        packed |= (unsigned)apply_post(p.po, q_clamp(q_mbqm(a, p.mult, p.shift) + p.zo, p.lo, p.hi), blockIdx.y, e0 + e) << (8 * e);
    }
    if ((((size_t)py) & 3) == 0) *(unsigned*)(py + e0) = packed;   // (a CONCATENATION part may start at any byte)
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) py[e0 + e] = (uint8_t)(packed >> (8 * e));
    }
}

// RESHAPE: a plain device copy as a kernel (a memcpy NODE in the captured plan crashed rocprofv3's
// kernel tracing on graph replay; kernels are also what the profiler can attribute)
__global__ __launch_bounds__(256) void tfl_copy_bytes(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n16 = n >> 4;
    if (t < n16) ((uint4*)y)[t] = ((const uint4*)x)[t];
    if (t < (n & 15)) y[(n16 << 4) + t] = x[(n16 << 4) + t];
}

struct AddQ { const uint8_t *a, *b; uint8_t* y; long long n; int za, zb, zo, m1, s1, m2, s2, mo, so, lo, hi; };
__global__ __launch_bounds__(256) void tfl_add_u8(const AddQ p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.n) return;
    const int v1 = q_mbqm(((int)p.a[t] - p.za) * (1 << 20), p.m1, p.s1);
    const int v2 = q_mbqm(((int)p.b[t] - p.zb) * (1 << 20), p.m2, p.s2);
    p.y[t] = (uint8_t)q_clamp(q_mbqm(v1 + v2, p.mo, p.so) + p.zo, p.lo, p.hi);
}

// requantise (QUANTIZE u8->u8, RELU/RELU6) : clamp(mbqm(q - zi) + zo, lo, hi)
__global__ __launch_bounds__(256) void tfl_requant_u8(const uint8_t* x, uint8_t* y, long long n, int zi, int zo, int m, int s, int lo, int hi) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) y[t] = (uint8_t)q_clamp(q_mbqm((int)x[t] - zi, m, s) + zo, lo, hi);
}
__global__ __launch_bounds__(256) void tfl_quantize_f32(const float* x, uint8_t* y, long long n, float scale, int zo) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float v = __fdiv_rn(x[t], scale);
    const float r = v >= 0.0f ? floorf(__fadd_rn(v, 0.5f)) : ceilf(__fsub_rn(v, 0.5f));
    y[t] = (uint8_t)q_clamp((int)r + zo, 0, 255);
}
__global__ __launch_bounds__(256) void tfl_dequantize_u8(const uint8_t* x, float* y, long long n, float scale, int z) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) y[t] = __fmul_rn(scale, (float)((int)x[t] - z));   // yolact.rs:177
}
__global__ __launch_bounds__(256) void tfl_lut_u8(const uint8_t* x, uint8_t* y, long long n, const uint8_t* lut) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) y[t] = lut[x[t]];
}
struct PadQ { const uint8_t* x; uint8_t* y; int id[4], od[4], before[4]; int fill; };
__global__ __launch_bounds__(256) void tfl_pad_u8(const PadQ p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)p.od[0] * p.od[1] * p.od[2] * p.od[3];
    if (t >= n) return;
    int c[4];
    long long r = t;
    for (int d = 3; d >= 0; --d) { c[d] = (int)(r % p.od[d]) - p.before[d]; r /= p.od[d]; }
    bool in = true;
    for (int d = 0; d < 4; ++d) in = in && (unsigned)c[d] < (unsigned)p.id[d];
    p.y[t] = in ? p.x[(((long long)c[0] * p.id[1] + c[1]) * p.id[2] + c[2]) * p.id[3] + c[3]] : (uint8_t)p.fill;
}
struct ResizeQ { const uint8_t* x; uint8_t* y; int H, W, C, Ho, Wo; float hs, ws; int half_pixel; long long ys; PostOps po; };
__global__ __launch_bounds__(256) void tfl_resize_bilinear_u8(const ResizeQ p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.Ho * p.Wo * p.C) return;
    const uint8_t* const px = p.x + (size_t)blockIdx.y * p.H * p.W * p.C;   // (batch plan: image-major)
    uint8_t* const py = p.y + blockIdx.y * p.ys;
    const int c = t % p.C, r0 = t / p.C, ox = r0 % p.Wo, oy = r0 / p.Wo;
    const float iy = p.half_pixel ? __fsub_rn(__fmul_rn(__fadd_rn((float)oy, 0.5f), p.hs), 0.5f) : __fmul_rn((float)oy, p.hs);
    const float ix = p.half_pixel ? __fsub_rn(__fmul_rn(__fadd_rn((float)ox, 0.5f), p.ws), 0.5f) : __fmul_rn((float)ox, p.ws);
    int y0 = (int)floorf(iy), y1 = (int)ceilf(iy), x0 = (int)floorf(ix), x1 = (int)ceilf(ix);
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > p.H - 1 ? p.H - 1 : y1; x1 = x1 > p.W - 1 ? p.W - 1 : x1;
    const float fy = __fsub_rn(iy, (float)y0), fx = __fsub_rn(ix, (float)x0);
    const float gy = __fsub_rn(1.0f, fy), gx = __fsub_rn(1.0f, fx);
    auto at = [&](int yy, int xx) { return (float)px[((size_t)yy * p.W + xx) * p.C + c]; };
    float v = __fmul_rn(__fmul_rn(at(y0, x0), gy), gx);
    v = __fadd_rn(v, __fmul_rn(__fmul_rn(at(y1, x0), fy), gx));
    v = __fadd_rn(v, __fmul_rn(__fmul_rn(at(y0, x1), gy), fx));
    v = __fadd_rn(v, __fmul_rn(__fmul_rn(at(y1, x1), fy), fx));
    const float r = floorf(__fadd_rn(v, 0.5f));
    py[t] = (uint8_t)apply_post(p.po, r < 0.0f ? 0 : (r > 255.0f ? 255 : (int)r), blockIdx.y, t);
}
// copy one concat input [outer][inner] into the output at column `off` of rows of `row` elements
struct CatQ { const uint8_t* x; uint8_t* y; long long outer; int inner, row, off, esz; int rescale; float sc, bias; int zo; };
__global__ __launch_bounds__(256) void tfl_concat_part(const CatQ p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.outer * p.inner * p.esz) return;
    const long long e = t / p.esz;
    const int b = (int)(t - e * p.esz);
    const long long o = e / p.inner;
    const int i = (int)(e - o * p.inner);
    const long long dst = ((o * p.row) + p.off + i) * p.esz + b;
    if (p.rescale) {
        const float v = __fadd_rn(__fmul_rn((float)p.x[t], p.sc), p.bias);
        const float r = v >= 0.0f ? floorf(__fadd_rn(v, 0.5f)) : ceilf(__fsub_rn(v, 0.5f));
        p.y[dst] = (uint8_t)q_clamp((int)r + p.zo, 0, 255);
    } else p.y[dst] = p.x[t];
}

// ---- host-side quantisation helpers (tflite::QuantizeMultiplier etc.)
void quantize_multiplier(double m, int* mult, int* shift) {
    if (m == 0.0) { *mult = 0; *shift = 0; return; }
    int e;
    const double q = frexp(m, &e);
    long long qf = (long long)floor(q * (double)(1ll << 31) + 0.5);
    if (qf == (1ll << 31)) { qf /= 2; ++e; }
    if (e < -31) { e = 0; qf = 0; }
    *mult = (int)qf; *shift = e;
}
void act_range(int act, float scale, int zp, int* lo, int* hi) {
    auto q = [&](float v) { const float d = v / scale; return zp + (int)(d >= 0 ? floorf(d + 0.5f) : ceilf(d - 0.5f)); };
    *lo = 0; *hi = 255;
    if (act == 1) { *lo = q(0.0f) > 0 ? q(0.0f) : 0; }
    else if (act == 3) { *lo = q(0.0f) > 0 ? q(0.0f) : 0; *hi = q(6.0f) < 255 ? q(6.0f) : 255; }
    else if (act == 2) { *lo = q(-1.0f) > 0 ? q(-1.0f) : 0; *hi = q(1.0f) < 255 ? q(1.0f) : 255; }
}
void same_pad(int in, int k, int stride, int dil, int* out, int* before) {
    *out = (in + stride - 1) / stride;
    const int eff = (k - 1) * dil + 1, total = (*out - 1) * stride + eff - in;
    *before = total > 0 ? total / 2 : 0;
}

enum PKind { P_CONV, P_DW, P_ADD, P_REQUANT, P_QUANT_F32, P_DEQUANT, P_LUT, P_PAD, P_RESIZE, P_CONCAT, P_COPY, P_CONV_I8 };
struct Prepared {
    PKind kind;
    int oi = -1;         // operator index in the model
    bool dead = false;   // folded into another launch (fuse_plan)
    int group = -1;           // >= 0: this convolution is launched as part of h->groups[group] (group_plan), by the group's first member
    ConvQ conv; ConvI8 ci8; AddQ add; PadQ pad; ResizeQ rs;
    std::vector<CatQ> cat;
    const void* src = nullptr; void* dst = nullptr; long long n = 0;
    int zi = 0, zo = 0, m = 0, s = 0, lo = 0, hi = 255; float scale = 1.0f;
    const uint8_t* lut = nullptr;
};

}  // namespace

struct yh_tfl {
    int dev = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<uint8_t> file;
    TflModel m;
    std::vector<void*> tens;
    std::vector<char> alias;    // tens[i] shares another tensor's buffer (RESHAPE): not freed twice
    std::vector<void*> extra;   // LUTs etc.
    std::vector<Prepared> plan;
    static constexpr int kMaxBatch = 2;          // images per invoke the buffers are sized for (the two tiles of yolact.rs:216-217)
    int nb = 1;                                  // images of the next invoke (yh_tfl_set_batch)
    bool batch_ok = true;                        // no operator of the model touches the image axis
    hipGraphExec_t gexecs[kMaxBatch] = { nullptr, nullptr };   // the plan per batch size, captured once and replayed (tensor addresses never change)
    // yh_tuning.tfl_group (round 4): independent register-fed convolutions of one kernel form and one depth of the plan's DAG as ONE
    // launch (group_plan); `order` is the plan in execution order (by depth; the identity when nothing is grouped)
    struct ConvGroup { std::vector<int> members; const ConvI8* probs[kMaxBatch] = { nullptr, nullptr }; const int* tile_start[kMaxBatch] = { nullptr, nullptr }; int tiles[kMaxBatch] = { 0, 0 }; };
    int use_group = 1;
    std::vector<ConvGroup> groups;
    std::vector<int> order;
    int use_fuse = 1;                 // yh_tuning.tfl_fuse: element-wise operators / PAD / CONCATENATION parts folded into their producers (fuse_plan)
    std::vector<char> gone;           // tensor i is never written by the fused plan (yh_tfl_tensor_read says so)
    int use_dot = 3, use_graph = 0;   // yh_tuning.tfl_dot (0 scalar kernel, 1 v_dot4 kernel, 2 + int8 MFMA kernel on LDS tiles where Ci % 64 == 0, 3 + its register-fed form - one wave per 16 x 16 tile - where Ci % 4 == 0: default) / tfl_graph
    hipStream_t side = nullptr;       // tfl_graph: carries the second branch of the captured graph
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* side_word = nullptr;
    // classify scratch
    uint32_t *frame_dev = nullptr, *codes_dev = nullptr, *stitch_dev = nullptr;
    uint8_t* tiles_dev = nullptr;
    float *rs_tmp = nullptr, *cells_dev = nullptr;
    int* diverged_dev = nullptr;
    size_t frame_cap = 0, rs_cap = 0;
    int fail(int code, const std::string& msg) { err = msg; return code; }
};

namespace {
thread_local std::string g_tfl_create_error;

#define TCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (h)->fail(YH_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

int prepare(yh_tfl* h) {
    TflModel& m = h->m;
    h->tens.assign(m.tensors.size(), nullptr);
    h->alias.assign(m.tensors.size(), 0);
    for (size_t i = 0; i < m.tensors.size(); ++i) {
        const TflTensor& t = m.tensors[i];
        const size_t bytes = t.count() * t.elem();
        if (bytes == 0) continue;
        const size_t alloc = t.data ? bytes : bytes * yh_tfl::kMaxBatch;   // activations: image-major, room for the batch plan
        TCHK(h, hipMalloc(&h->tens[i], alloc + 16));
        if (t.data) TCHK(h, hipMemcpy(h->tens[i], t.data, bytes, hipMemcpyHostToDevice));
        else TCHK(h, hipMemset(h->tens[i], 0, alloc));
    }
    auto T = [&](int i) -> const TflTensor& { return m.tensors[i]; };
    auto need = [&](bool c, const std::string& what) { if (!c && h->err.empty()) h->err = what; return c; };
    for (size_t oi = 0; oi < m.ops.size(); ++oi) {
        const TflOp& op = m.ops[oi];
        const std::string at = " (operator " + std::to_string(oi) + ")";
        Prepared pr = Prepared();   // (value-initialised: the kernel parameter blocks start zeroed - no folded operators)
        pr.oi = (int)oi;
        auto u8 = [&](int i) { return T(i).type == TFL_U8 && T(i).quant; };
        switch (op.code) {
            case TFL_CONV_2D:
            case TFL_DEPTHWISE_CONV_2D: {
                if (!need(op.in.size() >= 2 && op.out.size() == 1, "conv: bad arity" + at)) return YH_EINVAL;
                const TflTensor &x = T(op.in[0]), &w = T(op.in[1]), &y = T(op.out[0]);
                const bool dw = op.code == TFL_DEPTHWISE_CONV_2D;
                const int bi = op.in.size() > 2 ? op.in[2] : -1;
                if (!need(u8(op.in[0]) && u8(op.in[1]) && u8(op.out[0]) && x.shape.size() == 4 && w.shape.size() == 4 && y.shape.size() == 4 &&
                          x.shape[0] == 1 && (bi < 0 || (T(bi).type == TFL_I32 && T(bi).data)) && w.data,
                          "conv: only uint8 per-tensor quantised NHWC batch-1 convolutions with constant weights are supported" + at)) return YH_EINVAL;
                ConvQ& c = pr.conv;
                c.H = x.shape[1]; c.W = x.shape[2]; c.Ci = x.shape[3];
                c.kh = w.shape[1]; c.kw = w.shape[2];
                c.Co = dw ? w.shape[3] : w.shape[0];
                c.dm = dw ? op.depth_mult : 1;
                c.sh = op.stride_h; c.sw = op.stride_w; c.dh = op.dil_h; c.dw = op.dil_w;
                if (op.padding == 0) { same_pad(c.H, c.kh, c.sh, c.dh, &c.Ho, &c.ph); same_pad(c.W, c.kw, c.sw, c.dw, &c.Wo, &c.pw); }
                else { c.Ho = (c.H - ((c.kh - 1) * c.dh + 1) + c.sh) / c.sh; c.Wo = (c.W - ((c.kw - 1) * c.dw + 1) + c.sw) / c.sw; c.ph = c.pw = 0; }
                if (!need(y.shape[1] == c.Ho && y.shape[2] == c.Wo && y.shape[3] == c.Co && (dw ? (w.shape[0] == 1 && c.Co == c.Ci * c.dm) : w.shape[3] == c.Ci) &&
                          (bi < 0 || (int)T(bi).count() == c.Co), "conv: tensor shapes are inconsistent" + at)) return YH_EINVAL;
                c.x = (const uint8_t*)h->tens[op.in[0]]; c.w = (const uint8_t*)h->tens[op.in[1]];
                c.bias = bi >= 0 ? (const int*)h->tens[bi] : nullptr; c.y = (uint8_t*)h->tens[op.out[0]];
                c.zx = x.zp; c.zw = w.zp; c.zo = y.zp;
                c.xs = (long long)c.H * c.W * c.Ci; c.ys = (long long)c.Ho * c.Wo * c.Co;
                quantize_multiplier((double)x.scale * (double)w.scale / (double)y.scale, &c.mult, &c.shift);
                act_range(op.act, y.scale, y.zp, &c.lo, &c.hi);
                pr.kind = dw ? P_DW : P_CONV;
                c.wsum = nullptr;
                if (!dw && h->use_dot >= 2 && c.Ci % (h->use_dot >= 3 ? 4 : 64) == 0 && (((size_t)c.x) & 3) == 0 && w.data && (long long)c.kh * c.kw * c.Ci < 131072) {
                    // int8 MFMA form: the weight panel as int8 (w ^ 0x80) padded to 64-channel tiles, and per channel
                    //   cterm = (128 - zx) sum(w - 128) + K (128 - zx)(128 - zw) + bias   (all exact in int32: |.| < 2^31 for K < 2^17)
                    const int K = c.kh * c.kw * c.Ci, CoPad = (c.Co + 63) / 64 * 64;
                    std::vector<uint8_t> wq((size_t)CoPad * K + 16, 0x80);   // padding rows: w' = 0 (+ 16 bytes: the register-fed kernel's last load of a row may reach past it)
                    std::vector<int> ct(CoPad, 0);
                    const int* bias_h = bi >= 0 ? (const int*)T(bi).data : nullptr;
                    for (int o = 0; o < c.Co; ++o) {
                        long long sw = 0;
                        for (int k = 0; k < K; ++k) { const uint8_t b = w.data[(size_t)o * K + k]; wq[(size_t)o * K + k] = (uint8_t)(b ^ 0x80); sw += (int)b - 128; }
                        ct[o] = (int)((128 - c.zx) * sw + (long long)K * (128 - c.zx) * (128 - c.zw) + (bias_h ? bias_h[o] : 0));
                    }
                    void *dwq = nullptr, *dct = nullptr;
                    if (hipMalloc(&dwq, wq.size()) != hipSuccess || hipMalloc(&dct, ct.size() * 4) != hipSuccess) return h->fail(YH_ENOMEM, "hipMalloc int8 panel");
                    h->extra.push_back(dwq); h->extra.push_back(dct);
                    if (hipMemcpy(dwq, wq.data(), wq.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dct, ct.data(), ct.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                        return h->fail(YH_EHIP, "int8 panel upload");
                    ConvI8& q = pr.ci8;
                    q.x = c.x; q.wq = (const uint8_t*)dwq; q.cterm = (const int*)dct; q.y = c.y;
                    q.H = c.H; q.W = c.W; q.Ci = c.Ci; q.Ho = c.Ho; q.Wo = c.Wo; q.Co = c.Co; q.kh = c.kh; q.kw = c.kw; q.sh = c.sh; q.sw = c.sw;
                    q.ph = c.ph; q.pw = c.pw; q.dh = c.dh; q.dw = c.dw; q.zx = c.zx; q.zw = c.zw; q.zo = c.zo; q.mult = c.mult; q.shift = c.shift;
                    q.lo = c.lo; q.hi = c.hi; q.K = K; q.M = c.Ho * c.Wo; q.xs = c.xs; q.ys = c.ys;
                    pr.kind = P_CONV_I8;
                    break;
                }
                if (!dw && h->use_dot && c.Ci % 4 == 0 && w.data && (long long)c.kh * c.kw * c.Ci * 65025ll < (1ll << 31)) {
                    // per (channel, tap) sums of the raw weight bytes for the dot-product kernel
                    const int ntaps = c.kh * c.kw;
                    std::vector<int> ws((size_t)c.Co * ntaps, 0);
                    for (int o = 0; o < c.Co; ++o)
                        for (int tp = 0; tp < ntaps; ++tp) {
                            int sum = 0;
                            const uint8_t* wp = w.data + ((size_t)o * ntaps + tp) * c.Ci;
                            for (int ch = 0; ch < c.Ci; ++ch) sum += wp[ch];
                            ws[(size_t)o * ntaps + tp] = sum;
                        }
                    void* d = nullptr;
                    if (hipMalloc(&d, ws.size() * 4) != hipSuccess) return h->fail(YH_ENOMEM, "hipMalloc weight sums");
                    h->extra.push_back(d);
                    if (hipMemcpy(d, ws.data(), ws.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return h->fail(YH_EHIP, "weight sums upload");
                    c.wsum = (const int*)d;
                }
                break;
            }
            case TFL_ADD: {
                if (!need(op.in.size() == 2 && u8(op.in[0]) && u8(op.in[1]) && u8(op.out[0]) && T(op.in[0]).count() == T(op.out[0]).count() &&
                          T(op.in[1]).count() == T(op.out[0]).count(), "add: only same-shape uint8 tensors are supported" + at)) return YH_EINVAL;
                const TflTensor &a = T(op.in[0]), &b = T(op.in[1]), &y = T(op.out[0]);
                AddQ& q = pr.add;
                q.a = (const uint8_t*)h->tens[op.in[0]]; q.b = (const uint8_t*)h->tens[op.in[1]]; q.y = (uint8_t*)h->tens[op.out[0]];
                q.n = (long long)y.count(); q.za = a.zp; q.zb = b.zp; q.zo = y.zp;
                const double twice = 2.0 * (a.scale > b.scale ? (double)a.scale : (double)b.scale);
                quantize_multiplier((double)a.scale / twice, &q.m1, &q.s1);
                quantize_multiplier((double)b.scale / twice, &q.m2, &q.s2);
                quantize_multiplier(twice / ((double)(1 << 20) * (double)y.scale), &q.mo, &q.so);
                act_range(op.act, y.scale, y.zp, &q.lo, &q.hi);
                pr.kind = P_ADD;
                break;
            }
            case TFL_RELU:
            case TFL_RELU6:
            case TFL_QUANTIZE: {
                if (!need(op.in.size() == 1 && op.out.size() == 1 && u8(op.out[0]) && T(op.in[0]).count() == T(op.out[0]).count(), "quantize/relu: bad tensors" + at)) return YH_EINVAL;
                const TflTensor &x = T(op.in[0]), &y = T(op.out[0]);
                pr.src = h->tens[op.in[0]]; pr.dst = h->tens[op.out[0]]; pr.n = (long long)y.count();
                if (x.type == TFL_F32 && op.code == TFL_QUANTIZE) { pr.kind = P_QUANT_F32; pr.scale = y.scale; pr.zo = y.zp; break; }
                if (!need(u8(op.in[0]), "quantize/relu: input must be float32 or quantised uint8" + at)) return YH_EINVAL;
                pr.kind = P_REQUANT; pr.zi = x.zp; pr.zo = y.zp;
                quantize_multiplier((double)x.scale / (double)y.scale, &pr.m, &pr.s);
                act_range(op.code == TFL_RELU ? 1 : (op.code == TFL_RELU6 ? 3 : 0), y.scale, y.zp, &pr.lo, &pr.hi);
                break;
            }
            case TFL_DEQUANTIZE: {
                if (!need(op.in.size() == 1 && u8(op.in[0]) && T(op.out[0]).type == TFL_F32 && T(op.in[0]).count() == T(op.out[0]).count(), "dequantize: bad tensors" + at)) return YH_EINVAL;
                pr.kind = P_DEQUANT; pr.src = h->tens[op.in[0]]; pr.dst = h->tens[op.out[0]]; pr.n = (long long)T(op.out[0]).count();
                pr.scale = T(op.in[0]).scale; pr.zi = T(op.in[0]).zp;
                break;
            }
            case TFL_TANH: {
                if (!need(op.in.size() == 1 && u8(op.in[0]) && u8(op.out[0]) && T(op.in[0]).count() == T(op.out[0]).count(), "tanh: only uint8 is supported" + at)) return YH_EINVAL;
                const TflTensor &x = T(op.in[0]), &y = T(op.out[0]);
                uint8_t lut[256];
                const float inv = 1.0f / y.scale;
                for (int q = 0; q < 256; ++q) {   // PopulateLookupTable<uint8_t>
                    const float xv = x.scale * (float)(q - x.zp);
                    const float yv = (float)tanh((double)xv);
                    const float r = yv * inv;
                    const int rr = (int)(r >= 0 ? floorf(r + 0.5f) : ceilf(r - 0.5f)) + y.zp;
                    lut[q] = (uint8_t)(rr < 0 ? 0 : (rr > 255 ? 255 : rr));
                }
                void* d = nullptr;
                TCHK(h, hipMalloc(&d, 256));
                h->extra.push_back(d);
                TCHK(h, hipMemcpy(d, lut, 256, hipMemcpyHostToDevice));
                pr.kind = P_LUT; pr.lut = (const uint8_t*)d; pr.src = h->tens[op.in[0]]; pr.dst = h->tens[op.out[0]]; pr.n = (long long)y.count();
                break;
            }
            case TFL_PAD: {
                if (!need(op.in.size() == 2 && u8(op.in[0]) && u8(op.out[0]) && T(op.in[1]).type == TFL_I32 && T(op.in[1]).data &&
                          T(op.in[0]).shape.size() == 4 && T(op.in[1]).count() == 8, "pad: need a uint8 4-D input and constant [4,2] paddings" + at)) return YH_EINVAL;
                const TflTensor &x = T(op.in[0]), &y = T(op.out[0]);
                const int* pp = (const int*)T(op.in[1]).data;
                PadQ& q = pr.pad;
                for (int d = 0; d < 4; ++d) {
                    q.id[d] = x.shape[d]; q.before[d] = pp[2 * d]; q.od[d] = x.shape[d] + pp[2 * d] + pp[2 * d + 1];
                    if (!need(pp[2 * d] >= 0 && pp[2 * d + 1] >= 0 && y.shape[d] == q.od[d], "pad: output shape mismatch" + at)) return YH_EINVAL;
                }
                if (q.id[0] != 1 || q.od[0] != 1) h->batch_ok = false;
                q.x = (const uint8_t*)h->tens[op.in[0]]; q.y = (uint8_t*)h->tens[op.out[0]]; q.fill = y.zp;
                pr.kind = P_PAD;
                break;
            }
            case TFL_RESIZE_BILINEAR: {
                if (!need(op.in.size() == 2 && u8(op.in[0]) && u8(op.out[0]) && T(op.in[1]).type == TFL_I32 && T(op.in[1]).data && T(op.in[1]).count() == 2 &&
                          T(op.in[0]).shape.size() == 4 && T(op.in[0]).shape[0] == 1, "resize_bilinear: need uint8 NHWC batch 1 and a constant size" + at)) return YH_EINVAL;
                const TflTensor &x = T(op.in[0]), &y = T(op.out[0]);
                const int* sz = (const int*)T(op.in[1]).data;
                ResizeQ& q = pr.rs;
                q.H = x.shape[1]; q.W = x.shape[2]; q.C = x.shape[3]; q.Ho = sz[0]; q.Wo = sz[1]; q.half_pixel = op.half_pixel ? 1 : 0;
                if (!need(y.shape[1] == q.Ho && y.shape[2] == q.Wo && y.shape[3] == q.C, "resize_bilinear: output shape mismatch" + at)) return YH_EINVAL;
                q.hs = (op.align_corners && q.Ho > 1) ? (float)(q.H - 1) / (float)(q.Ho - 1) : (float)q.H / (float)q.Ho;
                q.ws = (op.align_corners && q.Wo > 1) ? (float)(q.W - 1) / (float)(q.Wo - 1) : (float)q.W / (float)q.Wo;
                q.x = (const uint8_t*)h->tens[op.in[0]]; q.y = (uint8_t*)h->tens[op.out[0]]; q.ys = (long long)q.Ho * q.Wo * q.C;
                pr.kind = P_RESIZE;
                break;
            }
            case TFL_CONCATENATION: {
                const TflTensor& y = T(op.out[0]);
                const int nd = (int)y.shape.size();
                const int axis = op.axis < 0 ? op.axis + nd : op.axis;
                if (!need(axis >= 0 && axis < nd && !op.in.empty(), "concatenation: bad axis" + at)) return YH_EINVAL;
                if (axis == 0) h->batch_ok = false;   // (joins along the image axis: this model runs one image per invoke only)
                long long outer = 1; int inner_o = 1;
                for (int d = 0; d < axis; ++d) outer *= y.shape[d];
                for (int d = axis; d < nd; ++d) inner_o *= y.shape[d];
                int off = 0;
                for (int ii : op.in) {
                    const TflTensor& x = T(ii);
                    if (!need(x.type == y.type && (int)x.shape.size() == nd, "concatenation: type/rank mismatch" + at)) return YH_EINVAL;
                    int inner = 1;
                    for (int d = axis; d < nd; ++d) inner *= x.shape[d];
                    CatQ c;
                    c.x = (const uint8_t*)h->tens[ii]; c.y = (uint8_t*)h->tens[op.out[0]]; c.outer = outer; c.inner = inner; c.row = inner_o; c.off = off;
                    c.esz = x.type == TFL_U8 ? 1 : 4; c.rescale = 0; c.sc = 1.0f; c.bias = 0.0f; c.zo = y.zp;
                    if (x.type == TFL_U8 && y.quant && (x.zp != y.zp || x.scale != y.scale)) {  // ConcatenationWithScaling
                        const float inv = 1.0f / y.scale;
                        c.rescale = 1; c.sc = x.scale * inv; c.bias = (float)(-x.zp) * c.sc;
                    }
                    pr.cat.push_back(c);
                    off += inner;
                }
                if (!need(off == inner_o, "concatenation: output shape mismatch" + at)) return YH_EINVAL;
                pr.kind = P_CONCAT;
                break;
            }
            case TFL_RESHAPE: {
                if (!need(!op.in.empty() && T(op.in[0]).count() * T(op.in[0]).elem() == T(op.out[0]).count() * T(op.out[0]).elem(), "reshape: size mismatch" + at)) return YH_EINVAL;
                // RESHAPE moves no bytes: every tensor is written once by its producer, so the output can
                // share the input's buffer (ops are in topological order: its consumers are prepared later
                // and pick the shared pointer up). Only a reshape of a constant keeps its own copy.
                if (!T(op.in[0]).data && !T(op.out[0]).data) {
                    hipFree(h->tens[op.out[0]]);
                    h->tens[op.out[0]] = h->tens[op.in[0]];
                    h->alias[op.out[0]] = 1;
                    continue;
                }
                pr.kind = P_COPY; pr.src = h->tens[op.in[0]]; pr.dst = h->tens[op.out[0]]; pr.n = (long long)(T(op.out[0]).count() * T(op.out[0]).elem());
                break;
            }
            default:
                h->err = op.code == TFL_CUSTOM
                             ? "custom operator '" + op.custom + "' (an EdgeTPU-compiled model? load the non-compiled FRC_model.tflite instead)" + at
                             : "unsupported builtin operator code " + std::to_string(op.code) + at;
                return YH_EINVAL;
        }
        h->plan.push_back(pr);
    }
    return YH_OK;
}

// Plan-time operator fusion (yh_tuning.tfl_fuse, default on). The reference runs 138 of its 141 operators as
// ONE compiled unit on the EdgeTPU (data/README.md:34-39); one launch per operator made ~70 small kernels at their ~5 us floor
// 0.35 ms of a 0.9 ms invoke. Folded here, all bit-exact by construction (PostOps above):
//   * QUANTIZE / RELU / RELU6 (uint8 -> uint8), TANH and ADD into the epilogue of the convolution, depthwise convolution or
//     RESIZE_BILINEAR that produces their operand last - when that tensor has no other reader and is not a graph output;
//   * a CONCATENATION part whose quantisation equals the output's and which is one contiguous block per image: its producer
//     writes straight into the concatenated tensor;
//   * PAD (height / width only) in front of convolutions: the convolution reads the unpadded tensor with the padding as its own.
// RESHAPE already shares its input's buffer. Tensors that are no longer written are listed in h->gone.
void fuse_plan(yh_tfl* h) {
    const TflModel& m = h->m;
    const int nt = (int)m.tensors.size(), nops = (int)m.ops.size();
    std::vector<int> root(nt), plan_of(nops, -1), out_refs(nt, 0), prod(nt, -1);
    for (int i = 0; i < nt; ++i) root[i] = i;
    for (const TflOp& op : m.ops)   // (topological order: the input's root is final when its reshape is met)
        if (op.code == TFL_RESHAPE && !op.in.empty() && h->tens[op.out[0]] == h->tens[op.in[0]]) root[op.out[0]] = root[op.in[0]];
    for (size_t pi = 0; pi < h->plan.size(); ++pi) plan_of[h->plan[pi].oi] = (int)pi;
    std::vector<std::vector<int>> cons(nt);
    for (int oi = 0; oi < nops; ++oi) {
        if (plan_of[oi] < 0) continue;   // (buffer-sharing RESHAPE: transparent)
        for (int t : m.ops[oi].in) if (t >= 0 && !m.tensors[t].data) cons[root[t]].push_back(oi);
        for (int t : m.ops[oi].out) prod[root[t]] = plan_of[oi];
    }
    for (int t : m.outputs) ++out_refs[root[t]];
    auto mark_gone = [&](int r) { for (int t = 0; t < nt; ++t) if (root[t] == r) h->gone[t] = 1; };
    auto is_conv = [](PKind k) { return k == P_CONV || k == P_CONV_I8 || k == P_DW; };
    // ---- PAD into the convolutions that read it
    for (Prepared& pd : h->plan) {
        if (pd.kind != P_PAD) continue;
        const PadQ& q = pd.pad;
        const int r = root[m.ops[pd.oi].out[0]];
        if (q.before[0] || q.before[3] || q.od[0] != q.id[0] || q.od[3] != q.id[3] || out_refs[r]) continue;
        bool all = !cons[r].empty();
        for (int c : cons[r]) all = all && is_conv(h->plan[plan_of[c]].kind) && root[m.ops[c].in[0]] == r;
        if (!all) continue;
        for (int c : cons[r]) {
            Prepared& cv = h->plan[plan_of[c]];
            ConvQ& k = cv.conv;
            k.x = q.x; k.H = q.id[1]; k.W = q.id[2]; k.ph += q.before[1]; k.pw += q.before[2]; k.xs = (long long)k.H * k.W * k.Ci;
            if (cv.kind == P_CONV_I8) { ConvI8& e = cv.ci8; e.x = k.x; e.H = k.H; e.W = k.W; e.ph = k.ph; e.pw = k.pw; e.xs = k.xs; }
        }
        pd.dead = true;
        mark_gone(r);
    }
    // ---- element-wise consumers, ADD and contiguous CONCATENATION parts into their producers
    for (size_t pi = 0; pi < h->plan.size(); ++pi) {
        Prepared& p = h->plan[pi];
        if (p.dead || !(is_conv(p.kind) || p.kind == P_RESIZE)) continue;
        PostOps po;
        memset(&po, 0, sizeof po);
        int cur = root[m.ops[p.oi].out[0]];
        uint8_t* y = nullptr;
        long long ys = 0;
        bool has_add = false;
        while (!out_refs[cur] && cons[cur].size() == 1) {
            const int c = cons[cur][0], cp = plan_of[c];
            if (cp <= (int)pi) break;
            Prepared& q = h->plan[cp];
            if (q.dead) break;
            if ((q.kind == P_REQUANT || q.kind == P_LUT) && po.n < 3) {
                PostStep& t = po.st[po.n++];
                t.kind = q.kind == P_REQUANT ? 1 : 2;
                t.zi = q.zi; t.zo = q.zo; t.m = q.m; t.s = q.s; t.lo = q.lo; t.hi = q.hi; t.lut = q.lut;
            } else if (q.kind == P_ADD && !has_add && po.n < 3) {
                const int a = root[m.ops[c].in[0]], b = root[m.ops[c].in[1]], other = a == cur ? b : a;
                // the other operand must be complete when this producer runs: written by an earlier launch (or the graph input / a constant)
                if (a == b || (prod[other] >= (int)pi) || h->gone[other]) break;
                po.st[po.n++].kind = 3;
                has_add = true;
                po.q_is_a = a == cur;
                po.other = po.q_is_a ? q.add.b : q.add.a;
                po.other_s = q.add.n;   // (uint8: elements = bytes per image)
                po.za = q.add.za; po.zb = q.add.zb; po.m1 = q.add.m1; po.s1 = q.add.s1; po.m2 = q.add.m2; po.s2 = q.add.s2;
                po.mo = q.add.mo; po.so = q.add.so; po.azo = q.add.zo; po.alo = q.add.lo; po.ahi = q.add.hi;
            } else if (q.kind == P_CONCAT) {
                size_t k = 0;
                for (; k < q.cat.size(); ++k) if (q.cat[k].x == (const uint8_t*)h->tens[cur]) break;
                if (k == q.cat.size()) break;
                const CatQ part = q.cat[k];
                if (part.rescale || part.esz != 1 || part.outer != 1) break;
                y = part.y + part.off; ys = part.row;
                q.cat.erase(q.cat.begin() + (long)k);
                if (q.cat.empty()) q.dead = true;
                mark_gone(cur);
                cur = -1;
                break;
            } else break;
            q.dead = true;
            mark_gone(cur);
            cur = root[m.ops[c].out[0]];
        }
        if (cur >= 0 && cur != root[m.ops[p.oi].out[0]]) { y = (uint8_t*)h->tens[cur]; h->gone[cur] = 0; for (int t = 0; t < nt; ++t) if (root[t] == cur) h->gone[t] = 0; }
        if (p.kind == P_RESIZE) { p.rs.po = po; if (y) { p.rs.y = y; if (ys) p.rs.ys = ys; } }
        else {
            p.conv.po = po; p.ci8.po = po;
            if (y) { p.conv.y = y; p.ci8.y = y; if (ys) { p.conv.ys = ys; p.ci8.ys = ys; } }
        }
    }
}

// What a plan entry reads and writes (device pointers into the tensors' allocations).
static void entry_io(const Prepared& p, std::vector<const void*>& rd, std::vector<const void*>& wr) {
    switch (p.kind) {
        case P_CONV: case P_DW: rd.push_back(p.conv.x); wr.push_back(p.conv.y); if (p.conv.po.other) rd.push_back(p.conv.po.other); break;
        case P_CONV_I8: rd.push_back(p.ci8.x); wr.push_back(p.ci8.y); if (p.ci8.po.other) rd.push_back(p.ci8.po.other); break;
        case P_ADD: rd.push_back(p.add.a); rd.push_back(p.add.b); wr.push_back(p.add.y); break;
        case P_PAD: rd.push_back(p.pad.x); wr.push_back(p.pad.y); break;
        case P_RESIZE: rd.push_back(p.rs.x); wr.push_back(p.rs.y); if (p.rs.po.other) rd.push_back(p.rs.po.other); break;
        case P_CONCAT: for (const CatQ& c : p.cat) { rd.push_back(c.x); wr.push_back(c.y); } break;
        default: rd.push_back(p.src); wr.push_back(p.dst); break;
    }
}

// yh_tuning.tfl_group: the plan is a DAG, and on this part a dependent launch costs 4.4 us however little it does - so independent
// convolutions that run the same register-fed kernel are launched TOGETHER. Every live entry gets its depth (longest chain of
// producers in front of it: an entry depends on the earlier writers of what it reads and, conservatively, on the earlier readers and
// writers of what it writes); the plan runs in depth order, which is a topological order; within a depth, the register-fed
// convolutions of one kernel form (same extent and k-steps) become one launch of tfl_conv_i8_direct_group with their parameter blocks in
// device memory. The 136-op model: the five levels' tower convolutions, their fifteen output convolutions and the FPN's three output
// convolutions are three launches instead of twenty-three. Same kernels on the same operands: same bytes.
int group_plan(yh_tfl* h) {
    struct Range { const char* lo; const char* hi; };
    std::vector<Range> al;
    for (size_t i = 0; i < h->tens.size(); ++i)
        if (h->tens[i] && !h->alias[i]) {
            const TflTensor& t = h->m.tensors[i];
            const size_t bytes = t.count() * t.elem();
            al.push_back(Range{ (const char*)h->tens[i], (const char*)h->tens[i] + (t.data ? bytes : bytes * yh_tfl::kMaxBatch) + 16 });
        }
    auto alloc_of = [&](const void* q) -> int {
        for (size_t i = 0; i < al.size(); ++i) if ((const char*)q >= al[i].lo && (const char*)q < al[i].hi) return (int)i;
        return -1;
    };
    const int np = (int)h->plan.size();
    std::vector<std::vector<int>> writers(al.size()), readers(al.size());
    std::vector<int> depth((size_t)np, 0);
    for (int i = 0; i < np; ++i) {
        const Prepared& p = h->plan[i];
        if (p.dead) continue;
        std::vector<const void*> rd, wr;
        entry_io(p, rd, wr);
        int dmax = 0;
        for (const void* q : rd) { const int a = alloc_of(q); if (a >= 0) for (int w : writers[a]) dmax = std::max(dmax, depth[w] + 1); }
        for (const void* q : wr) {
            const int a = alloc_of(q);
            if (a < 0) continue;
            for (int r : readers[a]) if (r != i) dmax = std::max(dmax, depth[r] + 1);
            // (writers of other PARTS of one allocation - CONCATENATION parts written in place - are independent of each other; a writer of the
            // same bytes is not)
            for (int w : writers[a]) {
                std::vector<const void*> r2, w2;
                entry_io(h->plan[w], r2, w2);
                for (const void* q2 : w2) if (q2 == q) dmax = std::max(dmax, depth[w] + 1);
            }
        }
        depth[i] = dmax;
        for (const void* q : rd) { const int a = alloc_of(q); if (a >= 0) readers[a].push_back(i); }
        for (const void* q : wr) { const int a = alloc_of(q); if (a >= 0) writers[a].push_back(i); }
    }
    h->order.clear();
    for (int i = 0; i < np; ++i) if (!h->plan[i].dead) h->order.push_back(i);
    std::stable_sort(h->order.begin(), h->order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
    // groups: consecutive runs of one depth in `order`, bucketed by kernel form
    auto candidate = [&](const Prepared& p) {
        if (p.kind != P_CONV_I8 || h->use_dot < 3) return false;
        ConvI8 q = p.ci8;
        q.M = q.Ho * q.Wo * yh_tfl::kMaxBatch;
        return conv_i8_direct_pays(q);
    };
    for (size_t a = 0; a < h->order.size();) {
        size_t b = a;
        while (b < h->order.size() && depth[h->order[b]] == depth[h->order[a]]) ++b;
        std::map<int, std::vector<int>> byform;
        for (size_t k = a; k < b; ++k) { const Prepared& p = h->plan[h->order[k]]; if (candidate(p)) byform[conv_i8_direct_form(p.ci8)].push_back(h->order[k]); }
        for (auto& kv : byform) {
            std::vector<int>& v = kv.second;
            for (size_t c0 = 0; c0 + 1 < v.size(); c0 += kMaxGroup) {
                const size_t c1 = std::min(v.size(), c0 + (size_t)kMaxGroup);
                if (c1 - c0 < 2) break;
                yh_tfl::ConvGroup g;
                g.members.assign(v.begin() + c0, v.begin() + c1);
                for (int nb = 1; nb <= yh_tfl::kMaxBatch; ++nb) {
                    std::vector<ConvI8> probs;
                    std::vector<int> start(1, 0);
                    for (int mi : g.members) {
                        ConvI8 q = h->plan[mi].ci8;
                        q.M = q.Ho * q.Wo * nb;
                        probs.push_back(q);
                        start.push_back(start.back() + ((q.M + 15) / 16) * ((q.Co + 15) / 16));
                    }
                    void *dp = nullptr, *ds = nullptr;
                    if (hipMalloc(&dp, probs.size() * sizeof(ConvI8)) != hipSuccess || hipMalloc(&ds, start.size() * sizeof(int)) != hipSuccess) return h->fail(YH_ENOMEM, "hipMalloc conv group");
                    h->extra.push_back(dp); h->extra.push_back(ds);
                    if (hipMemcpy(dp, probs.data(), probs.size() * sizeof(ConvI8), hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(ds, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return h->fail(YH_EHIP, "conv group upload");
                    g.probs[nb - 1] = (const ConvI8*)dp; g.tile_start[nb - 1] = (const int*)ds; g.tiles[nb - 1] = start.back();
                }
                for (int mi : g.members) h->plan[mi].group = (int)h->groups.size();
                h->groups.push_back(g);
            }
        }
        a = b;
    }
    return YH_OK;
}


int enqueue_plan(yh_tfl* h) {
    const hipStream_t s = h->stream;
    const unsigned nb = (unsigned)h->nb;   // images of this invoke: activations are image-major, so element-wise ops just see nb x the elements
    TraceRange tr_all("yh_tfl:plan(enqueue)");
    for (size_t oi = 0; oi < h->order.size(); ++oi) {
        const size_t pi = (size_t)h->order[oi];   // (execution order: by depth of the plan's DAG where convolutions are grouped, else the file's)
        const Prepared& p = h->plan[pi];
        if (p.dead) continue;   // folded into another launch (fuse_plan)
        if (p.group >= 0 && h->groups[p.group].members[0] != (int)pi) continue;   // launched with its group's first member
        static const char* kind_name[] = { "CONV_2D", "DEPTHWISE_CONV_2D", "ADD", "RELU/QUANTIZE", "QUANTIZE(f32)", "DEQUANTIZE", "TANH", "PAD", "RESIZE_BILINEAR", "CONCATENATION", "RESHAPE(copy)", "CONV_2D(int8 MFMA)" };
        TraceRange tr(kind_name[p.kind]);   // (roctx: one range per operator, named by its TFLite op; a no-op unless a tracer is attached)
        switch (p.kind) {
            case P_CONV:
                if (p.conv.wsum) {
                    const dim3 grid((unsigned)(((long long)p.conv.Ho * p.conv.Wo + 63) / 64), (unsigned)((p.conv.Co + 7) / 8), nb);
                    if (p.conv.Ci % 16 == 0) hipLaunchKernelGGL(tfl_conv_u8_dot<4>, grid, dim3(256), 0, s, p.conv);
                    else hipLaunchKernelGGL(tfl_conv_u8_dot<1>, grid, dim3(64), 0, s, p.conv);
                }
                else if (h->use_dot && p.conv.kh * p.conv.kw * p.conv.Ci <= kPx8MaxK) {
                    const dim3 grid((unsigned)(((long long)p.conv.Ho * p.conv.Wo + 255) / 256), (unsigned)((p.conv.Co + 7) / 8), nb);
                    hipLaunchKernelGGL((tfl_conv_u8_px8<0, 0, 0>), grid, dim3(256), 0, s, p.conv);   // (the <3, 3, 3> form - all 27 input bytes asked for up front - measured 13.0 us against 10.6)
                }
                else hipLaunchKernelGGL(tfl_conv_u8, dim3(nblk((long long)p.conv.Ho * p.conv.Wo * p.conv.Co), nb), dim3(256), 0, s, p.conv);
                break;
            case P_CONV_I8: {
                ConvI8 q = p.ci8;
                q.M = q.Ho * q.Wo * (int)nb;
                if (p.group >= 0) {
                    const yh_tfl::ConvGroup& g = h->groups[p.group];
                    launch_conv_i8_direct(DirectLaunch{ &q, g.probs[nb - 1], g.tile_start[nb - 1], (int)g.members.size(), g.tiles[nb - 1] }, s);
                }
                else if (h->use_dot >= 3 && conv_i8_direct_pays(q)) launch_conv_i8_direct(DirectLaunch{ &q, nullptr, nullptr, 0, 0 }, s);
                else hipLaunchKernelGGL(tfl_conv_i8_mfma, dim3((unsigned)((q.M + 63) / 64), (unsigned)((q.Co + 63) / 64)), dim3(256), 0, s, q);
                break;
            }
            case P_DW:
                if (h->use_dot && p.conv.dm == 1 && p.conv.kh == 3 && p.conv.kw == 3 && p.conv.Co % 4 == 0 && (((size_t)p.conv.x | (size_t)p.conv.w) & 3) == 0)
                    hipLaunchKernelGGL((tfl_dwconv_u8_c4<3, 3>), dim3(nblk((long long)p.conv.Ho * p.conv.Wo * (p.conv.Co >> 2)), nb), dim3(256), 0, s, p.conv);
                else hipLaunchKernelGGL(tfl_dwconv_u8, dim3(nblk((long long)p.conv.Ho * p.conv.Wo * p.conv.Co), nb), dim3(256), 0, s, p.conv);
                break;
            case P_ADD: { AddQ q = p.add; q.n *= nb; hipLaunchKernelGGL(tfl_add_u8, dim3(nblk(q.n)), dim3(256), 0, s, q); break; }
            case P_REQUANT: hipLaunchKernelGGL(tfl_requant_u8, dim3(nblk(p.n * nb)), dim3(256), 0, s, (const uint8_t*)p.src, (uint8_t*)p.dst, p.n * nb, p.zi, p.zo, p.m, p.s, p.lo, p.hi); break;
            case P_QUANT_F32: hipLaunchKernelGGL(tfl_quantize_f32, dim3(nblk(p.n * nb)), dim3(256), 0, s, (const float*)p.src, (uint8_t*)p.dst, p.n * nb, p.scale, p.zo); break;
            case P_DEQUANT: hipLaunchKernelGGL(tfl_dequantize_u8, dim3(nblk(p.n * nb)), dim3(256), 0, s, (const uint8_t*)p.src, (float*)p.dst, p.n * nb, p.scale, p.zi); break;
            case P_LUT: hipLaunchKernelGGL(tfl_lut_u8, dim3(nblk(p.n * nb)), dim3(256), 0, s, (const uint8_t*)p.src, (uint8_t*)p.dst, p.n * nb, p.lut); break;
            case P_PAD: { PadQ q = p.pad; q.id[0] *= (int)nb; q.od[0] *= (int)nb; hipLaunchKernelGGL(tfl_pad_u8, dim3(nblk((long long)q.od[0] * q.od[1] * q.od[2] * q.od[3])), dim3(256), 0, s, q); break; }
            case P_RESIZE: hipLaunchKernelGGL(tfl_resize_bilinear_u8, dim3(nblk((long long)p.rs.Ho * p.rs.Wo * p.rs.C), nb), dim3(256), 0, s, p.rs); break;
            case P_CONCAT: for (CatQ c : p.cat) { c.outer *= nb; hipLaunchKernelGGL(tfl_concat_part, dim3(nblk(c.outer * c.inner * c.esz)), dim3(256), 0, s, c); } break;
            case P_COPY: hipLaunchKernelGGL(tfl_copy_bytes, dim3(nblk(((p.n * nb) >> 4) + 16)), dim3(256), 0, s, (const uint8_t*)p.src, (uint8_t*)p.dst, p.n * nb); break;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return h->fail(YH_EHIP, std::string("tflite plan launch: ") + hipGetErrorString(e));
    return YH_OK;
}

// yh_tuning.tfl_graph: 0 eager launches (default), 1 the plan as one captured hipGraph (with a 4-byte memset captured on
// a side stream beside it, so that the graph has two branches). Per invoke on the 136-op model (set_input + invoke + read
// output 4, median of 300): eager 1.104 ms, two-branch graph 1.222 ms, single-branch graph 1.096 ms - the step is bound by
// kernel time, not by launches, and the one graph form that is not slower is the one that cannot ship:
// round 1 shipped eager launches because `rocprofv3 --kernel-trace` crashed inside hipGraphLaunch on the graph. The cause
// (round 2, profiles/r02_graph_replay_under_rocprofv3.md with the logs) is not the plan: the HIP runtime replays a
// SINGLE-BRANCH graph from AQL packets it pre-built at instantiation, and rocprofv3's queue interception faults on that
// path - the YOLACT engine's own step crashes the same way when captured without a second branch, every forked capture is
// fine, and so is the single-branch form with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0. The same replay path later produced a GPU
// memory access fault WITHOUT the profiler (engine.hip, enqueue_all), so this library never builds a single-branch graph.
int run_plan(yh_tfl* h) {
    if (!h->use_graph) return enqueue_plan(h);
    hipGraphExec_t& gexec = h->gexecs[h->nb - 1];
    if (!gexec) {
        hipGraph_t g = nullptr;
        TCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
        // Inside the capture window nothing returns early: a failure is collected, the side stream is joined and the capture
        // is ENDED in every case (a stream left capturing would fail every later call on the handle).
        hipError_t ce = hipSuccess;
        const char* where = "";
        auto step = [&](hipError_t e, const char* what) { if (ce == hipSuccess && e != hipSuccess) { ce = e; where = what; } };
        // a second branch: the runtime then replays the graph node by node (see above)
        step(hipEventRecord(h->ev_fork, h->stream), "fork record");
        step(hipStreamWaitEvent(h->side, h->ev_fork, 0), "fork wait");
        step(launch_side_touch((unsigned*)h->side_word, h->side), "side branch");   // (a kernel node: no memset node in any capture)
        step(hipEventRecord(h->ev_join, h->side), "join record");
        const int rc = ce == hipSuccess ? enqueue_plan(h) : YH_OK;
        step(hipStreamWaitEvent(h->stream, h->ev_join, 0), "join wait");   // (also after a failed enqueue: an unjoined fork invalidates the capture)
        const hipError_t e = hipStreamEndCapture(h->stream, &g);
        if (rc || ce != hipSuccess || e != hipSuccess || !g) {
            if (g) hipGraphDestroy(g);
            (void)hipGetLastError();
            if (rc) return rc;   // (enqueue_plan has set the message)
            return h->fail(YH_EHIP, ce != hipSuccess ? std::string("tflite plan capture (") + where + "): " + hipGetErrorString(ce)
                                                     : std::string("tflite plan capture: ") + hipGetErrorString(e));
        }
        const hipError_t ei = hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (ei != hipSuccess) { gexec = nullptr; return h->fail(YH_EHIP, std::string("tflite plan instantiate: ") + hipGetErrorString(ei)); }
    }
    TCHK(h, hipGraphLaunch(gexec, h->stream));
    return YH_OK;
}

void fill_info(const TflTensor& t, yh_tensor_info* info) {
    info->name = t.name.c_str();
    info->kind = t.type == TFL_U8 ? YH_KIND_U8 : (t.type == TFL_F32 ? YH_KIND_F32 : t.type);
    info->ndims = (int)(t.shape.size() < 4 ? t.shape.size() : 4);
    for (int d = 0; d < 4; ++d) info->dims[d] = d < (int)t.shape.size() ? t.shape[d] : 1;
    info->scale = t.scale; info->zero_point = t.zp;
}

}  // namespace

extern "C" {

int yh_tfl_validate(const void* model_bytes, size_t nbytes, int32_t* n_tensors, int32_t* n_ops, char* err, size_t err_cap) {
    if (!model_bytes) return YH_EINVAL;
    TflModel m;
    const bool ok = m.parse((const uint8_t*)model_bytes, nbytes);
    if (n_tensors) *n_tensors = (int32_t)m.tensors.size();
    if (n_ops) *n_ops = (int32_t)m.ops.size();
    if (err && err_cap) { strncpy(err, m.error.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return ok ? YH_OK : YH_EWEIGHTS;
}

const char* yh_tfl_last_error(const yh_tfl* h) { return h ? h->err.c_str() : g_tfl_create_error.c_str(); }

int yh_tfl_create(const void* model_bytes, size_t nbytes, int32_t device, yh_tfl** out) {
    return yh_tfl_create_tuned(model_bytes, nbytes, device, nullptr, out);
}

int yh_tfl_create_tuned(const void* model_bytes, size_t nbytes, int32_t device, const yh_tuning* tune, yh_tfl** out) {
    if (!model_bytes || !out) { g_tfl_create_error = "null argument"; return YH_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_tfl_create_error = "no such HIP device (no CPU fallback)"; return YH_EHIP; }
    yh_tfl* h = new yh_tfl();
    h->dev = device;
    if (tune && tune->tfl_dot >= 0) h->use_dot = tune->tfl_dot;
    if (tune && tune->tfl_graph >= 0) h->use_graph = tune->tfl_graph;
    if (tune && tune->tfl_fuse >= 0) h->use_fuse = tune->tfl_fuse;
    if (tune && tune->tfl_group >= 0) h->use_group = tune->tfl_group;
    h->file.assign((const uint8_t*)model_bytes, (const uint8_t*)model_bytes + nbytes);
    auto bail = [&](int rc) { g_tfl_create_error = h->err; yh_tfl_destroy(h); return rc; };
    if (!h->m.parse(h->file.data(), h->file.size())) { h->err = "tflite parse: " + h->m.error; return bail(YH_EWEIGHTS); }
    if (h->m.inputs.size() != 1) { h->err = "expected exactly one graph input"; return bail(YH_EINVAL); }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess || hipMalloc(&h->side_word, 16) != hipSuccess) { h->err = "device setup failed"; return bail(YH_EHIP); }
    int rc = prepare(h);
    if (rc) return bail(rc);
    h->gone.assign(h->m.tensors.size(), 0);
    if (h->use_fuse) fuse_plan(h);
    for (int i = 0; i < (int)h->plan.size(); ++i) h->order.push_back(i);
    if (h->use_group && (rc = group_plan(h))) return bail(rc);
    *out = h;
    return YH_OK;
}

// Launches per invoke of the plan, and how many of them are CONV_2D / of those on the int8 matrix pipes.
int yh_tfl_plan_info(const yh_tfl* h, int32_t* launches, int32_t* conv_launches, int32_t* conv_mfma_launches) {
    if (!h) return YH_EINVAL;
    int n = 0, nc = 0, nm = 0;
    for (size_t i = 0; i < h->plan.size(); ++i) {
        const Prepared& p = h->plan[i];
        if (p.dead) continue;
        if (p.group >= 0 && h->groups[p.group].members[0] != (int)i) { ++nc; ++nm; continue; }   // (a convolution launched with its group)
        n += p.kind == P_CONCAT ? (int)p.cat.size() : 1;
        if (p.kind == P_CONV || p.kind == P_CONV_I8) ++nc;
        if (p.kind == P_CONV_I8) ++nm;
    }
    if (launches) *launches = n;
    if (conv_launches) *conv_launches = nc;
    if (conv_mfma_launches) *conv_mfma_launches = nm;
    return YH_OK;
}

void yh_tfl_destroy(yh_tfl* h) {
    if (!h) return;
    hipSetDevice(h->dev);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (hipGraphExec_t ge : h->gexecs) if (ge) hipGraphExecDestroy(ge);
    if (h->side) { hipStreamSynchronize(h->side); hipStreamDestroy(h->side); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side_word) hipFree(h->side_word);
    for (size_t i = 0; i < h->tens.size(); ++i) if (h->tens[i] && !(i < h->alias.size() && h->alias[i])) hipFree(h->tens[i]);
    for (void* p : h->extra) hipFree(p);
    void* scratch[] = { h->frame_dev, h->codes_dev, h->stitch_dev, h->tiles_dev, h->rs_tmp, h->cells_dev, h->diverged_dev };
    for (void* p : scratch) if (p) hipFree(p);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int yh_tfl_input_info(const yh_tfl* h, yh_tensor_info* info) {
    if (!h || !info) return YH_EINVAL;
    fill_info(h->m.tensors[h->m.inputs[0]], info);
    return YH_OK;
}
int yh_tfl_output_count(const yh_tfl* h) { return h ? (int)h->m.outputs.size() : YH_EINVAL; }
int yh_tfl_output_info(const yh_tfl* h, int32_t i, yh_tensor_info* info) {
    if (!h || !info || i < 0 || i >= (int)h->m.outputs.size()) return YH_EINVAL;
    fill_info(h->m.tensors[h->m.outputs[i]], info);
    return YH_OK;
}
int yh_tfl_tensor_count(const yh_tfl* h) { return h ? (int)h->m.tensors.size() : YH_EINVAL; }

int yh_tfl_set_batch(yh_tfl* h, int32_t n_images) {
    if (!h) return YH_EINVAL;
    if (n_images < 1 || n_images > yh_tfl::kMaxBatch) return h->fail(YH_EINVAL, "1 or 2 images per invoke");
    if (n_images > 1 && !h->batch_ok) return h->fail(YH_EINVAL, "an operator of this model works along the image axis: one image per invoke only");
    h->nb = n_images;
    return YH_OK;
}

int yh_tfl_set_input(yh_tfl* h, const void* data, size_t nbytes) {
    if (!h || !data) return YH_EINVAL;
    const TflTensor& t = h->m.tensors[h->m.inputs[0]];
    if (nbytes != t.count() * t.elem() * (size_t)h->nb) return h->fail(YH_EINVAL, "input size mismatch");   // the reference only warns (yolact.rs:151-158)
    TCHK(h, hipSetDevice(h->dev));
    TCHK(h, hipMemcpyAsync(h->tens[h->m.inputs[0]], data, nbytes, hipMemcpyHostToDevice, h->stream));
    // copy_from_slice semantics (yolact.rs:161-162): the caller's buffer is free again on return. From pageable memory the
    // runtime has staged the bytes by now; from pinned / registered memory the DMA is still reading: wait for it.
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, data) == hipSuccess && at.type == hipMemoryTypeHost) TCHK(h, hipStreamSynchronize(h->stream));
    else (void)hipGetLastError();
    return YH_OK;
}
int yh_tfl_invoke(yh_tfl* h) {
    if (!h) return YH_EINVAL;
    TCHK(h, hipSetDevice(h->dev));
    TraceRange tr("yh_tfl_invoke");
    return run_plan(h);
}
int yh_tfl_tensor_read(yh_tfl* h, int32_t tensor, void* dst, size_t nbytes) {
    if (!h || !dst || tensor < 0 || tensor >= (int)h->m.tensors.size()) return YH_EINVAL;
    const TflTensor& t = h->m.tensors[tensor];
    if (tensor < (int)h->gone.size() && h->gone[tensor])
        return h->fail(YH_ESTATE, "tensor '" + t.name + "' is folded into its producer's launch and never written (yh_tuning.tfl_fuse = 0 materialises every tensor)");
    if (nbytes != t.count() * t.elem() * (size_t)(t.data ? 1 : h->nb)) return h->fail(YH_EINVAL, "tensor size mismatch (activations hold yh_tfl_set_batch images)");
    TCHK(h, hipSetDevice(h->dev));
    TCHK(h, hipMemcpyAsync(dst, h->tens[tensor], nbytes, hipMemcpyDeviceToHost, h->stream));
    TCHK(h, hipStreamSynchronize(h->stream));
    return YH_OK;
}
int yh_tfl_output_read(yh_tfl* h, int32_t i, void* dst, size_t nbytes) {
    if (!h || i < 0 || i >= (int)h->m.outputs.size()) return YH_EINVAL;
    return yh_tfl_tensor_read(h, h->m.outputs[i], dst, nbytes);
}

int yh_tfl_classify_frame_u32(yh_tfl* h, uint32_t* frame, int32_t w, int32_t hh, int32_t mode) {
    if (!h || !frame || w < 1 || hh < 1) return YH_EINVAL;
    if (mode != YH_COMPAT_STRICT && mode != YH_COMPAT_SANE) return h->fail(YH_EINVAL, "bad compat mode");
    const TflTensor& in = h->m.tensors[h->m.inputs[0]];
    if (in.type != TFL_U8 || in.shape.size() != 4 || in.shape[0] != 1 || in.shape[1] != in.shape[2] || in.shape[3] != 3 || in.shape[1] % 8 != 0 || in.shape[1] / 8 > 64)
        return h->fail(YH_EINVAL, "classify needs a [1,S,S,3] uint8 input with S % 8 == 0");
    if (h->m.outputs.size() < 5) return h->fail(YH_EINVAL, "classify reads output index 4 (yolact.rs:91): the model has fewer outputs");
    const TflTensor& o4 = h->m.tensors[h->m.outputs[4]];
    const int S = in.shape[1], grid = S / 8;
    const size_t cells = (size_t)grid * grid;
    if (o4.count() % cells != 0 || o4.count() / cells < 4 || (o4.type != TFL_U8 && o4.type != TFL_F32))
        return h->fail(YH_EINVAL, "output 4 must hold (S/8)^2 cells of >= 4 uint8 or float32 logits");
    const int C = (int)(o4.count() / cells);
    TCHK(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)w * hh;
    auto grow = [&](void** p, size_t* cap, size_t bytes) { if (bytes <= *cap) return true; if (*p) hipFree(*p); *p = nullptr; *cap = 0; if (hipMalloc(p, bytes) != hipSuccess) return false; *cap = bytes; return true; };
    const size_t tmp_need = (size_t)12 * ((size_t)w * S > (size_t)2 * S * hh ? (size_t)w * S : (size_t)2 * S * hh);
    if (!grow((void**)&h->frame_dev, &h->frame_cap, npx * 4) || !grow((void**)&h->rs_tmp, &h->rs_cap, tmp_need)) return h->fail(YH_ENOMEM, "hipMalloc scratch");
    if (!h->tiles_dev) {
        if (hipMalloc((void**)&h->tiles_dev, (size_t)2 * S * S * 3) != hipSuccess || hipMalloc((void**)&h->cells_dev, 2 * cells * C * 4) != hipSuccess ||
            hipMalloc((void**)&h->codes_dev, 2 * cells * 4) != hipSuccess || hipMalloc((void**)&h->stitch_dev, (size_t)2 * S * S * 4) != hipSuccess ||
            hipMalloc((void**)&h->diverged_dev, 8) != hipSuccess) return h->fail(YH_ENOMEM, "hipMalloc scratch");
    }
    hipStream_t s = h->stream;
    TraceRange tr("yh_tfl_classify_frame_u32");
    // yolact.rs:195-214
    TCHK(h, hipMemcpyAsync(h->frame_dev, frame, npx * 4, hipMemcpyHostToDevice, s));
    hipError_t e = launch_resize_v_u32(h->frame_dev, w, hh, h->rs_tmp, S, s);
    const int nb_saved = h->nb;
    if (h->batch_ok) {
        // yolact.rs:216-217: the two tiles are independent - they run as ONE invoke of the batch plan (the horizontal resize
        // writes them, image-major, straight into the input tensor): half the launches per frame, twice the work per launch
        if (e == hipSuccess) e = launch_resize_h(h->rs_tmp, w, S, h->tens[h->m.inputs[0]], 2 * S, 1, s);
        if (e != hipSuccess) return h->fail(YH_EHIP, "classify pre failed");
        h->nb = 2;
        const int rc = run_plan(h);
        h->nb = nb_saved;
        if (rc) return rc;
        // yolact.rs:169-182: outputs -> f32 (results[4]), both tiles
        if (o4.type == TFL_U8) hipLaunchKernelGGL(tfl_dequantize_u8, dim3(nblk((long long)2 * cells * C)), dim3(256), 0, s, (const uint8_t*)h->tens[h->m.outputs[4]], h->cells_dev, (long long)(2 * cells * C), o4.scale, o4.zp);
        else TCHK(h, hipMemcpyAsync(h->cells_dev, h->tens[h->m.outputs[4]], 2 * cells * C * 4, hipMemcpyDeviceToDevice, s));
    } else {
        if (e == hipSuccess) e = launch_resize_h(h->rs_tmp, w, S, h->tiles_dev, 2 * S, 1, s);
        if (e != hipSuccess) return h->fail(YH_EHIP, "classify pre failed");
        h->nb = 1;
        for (int t = 0; t < 2; ++t) {   // one invoke per tile (an operator of this model works along the image axis)
            TCHK(h, hipMemcpyAsync(h->tens[h->m.inputs[0]], h->tiles_dev + (size_t)t * S * S * 3, (size_t)S * S * 3, hipMemcpyDeviceToDevice, s));
            const int rc = run_plan(h);
            if (rc) { h->nb = nb_saved; return rc; }
            float* dst = h->cells_dev + (size_t)t * cells * C;
            if (o4.type == TFL_U8) hipLaunchKernelGGL(tfl_dequantize_u8, dim3(nblk((long long)cells * C)), dim3(256), 0, s, (const uint8_t*)h->tens[h->m.outputs[4]], dst, (long long)(cells * C), o4.scale, o4.zp);
            else TCHK(h, hipMemcpyAsync(dst, h->tens[h->m.outputs[4]], cells * C * 4, hipMemcpyDeviceToDevice, s));
        }
        h->nb = nb_saved;
    }
    e = launch_cells_postprocess(h->cells_dev, 2, grid, C, mode, h->codes_dev, h->diverged_dev, s);   // yolact.rs:90-131
    if (e == hipSuccess) e = launch_upsample_codes(h->codes_dev, 2, grid, h->stitch_dev, 1, s);        // :127-128, :219-220
    if (e == hipSuccess) e = launch_resize_v_u32(h->stitch_dev, 2 * S, S, h->rs_tmp, hh, s);            // :222-231
    if (e == hipSuccess) e = launch_resize_h(h->rs_tmp, 2 * S, hh, h->frame_dev, w, 2, s);
    if (e != hipSuccess) return h->fail(YH_EHIP, "classify post failed");
    int div[2] = { 0, 0 };
    TCHK(h, hipMemcpyAsync(div, h->diverged_dev, 8, hipMemcpyDeviceToHost, s));
    TCHK(h, hipStreamSynchronize(s));
    if (div[0] || div[1]) return h->fail(YH_EDIVERGE, "reference flood fill (yolact.rs:57-78) does not terminate on this frame");
    TCHK(h, hipMemcpy(frame, h->frame_dev, npx * 4, hipMemcpyDeviceToHost));
    return YH_OK;
}

}  // extern "C"
