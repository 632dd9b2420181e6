// conv_direct_f16 - small convolutions (few output pixels, long K: ResNet layers 3-4 at batch 1-4) with NO LDS staging, no
// barrier in the K loop and no split-K launch pair.
//
// What the LDS-tiled kernels (conv_igemm.hip) cost at batch 1 is not their MFMAs: a layer-3 3x3 convolution (1225 pixels x 256
// channels, K = 2304) has 80 tiles of 64 x 64, so it is split six ways along K, every slice writes a 16 KB f32 slab and a second
// launch sums the slabs - 11 us + 6.5 us for 1.45 GFLOP, on a chip where no launch takes less than 4.4 us. The TFLite executor's
// int8 convolutions had the same shape and went from 13.5 to 6 us per launch by feeding the MFMAs straight from global memory
// (tflite_exec.hip, tfl_conv_i8_direct); this is the f16 form of that kernel.
//
// A tile = 32 output channels x 32 output pixels of v_mfma_f32_32x32x16_f16, whose operand registers are plain 16-byte buffer
// loads: lane (l31, g) holds K elements 8 g .. 8 g + 7 of row l31, contiguous both in the weight panel [Co][(r, s, c)] and in an
// NHWC pixel. KS = 4: a workgroup of four waves, every wave the WHOLE tile for a quarter of K (of each 64-channel chunk of the
// input, wave w takes channels 16 w .. 16 w + 15: the four waves together read one 128-byte line of a row), the four partial
// tiles meet once in LDS; KS = 1 (short K): one wave per tile, no LDS at all. The K loop is as bare as the hardware allows: the tap
// of a step is a compile-time constant (KK = 1 or 3: an iteration of a 3x3 launch is the nine taps of one 16-channel slice), so
// every per-lane address is computed ONCE - one byte offset per tap for the pixel (a tap outside the image, or a row past M, gets
// an offset past the buffer: the load returns zeros, there is no select), one for the weight row - and a step is two buffer loads
// whose scalar offset carries the channel slice, one MFMA and the scalar adds. (The first version computed positions and selects
// per step: 63 instructions per MFMA, 22 us for a launch the tiled kernel does in 7.)
// NR rings of D steps each rotate: while iteration `it` is consumed out of one ring, iteration it + NR - 1 is loaded into the ring
// that was consumed last - NR - 1 iterations of loads are in flight per wave (NR = 3, D = 9: 36 loads of 1 KB). Distinct rings, not
// one ring refilled slot by slot: that form made the register allocator rotate the ring with v_mov copies at the top of the loop,
// and a copy of a register that was just loaded is a wait for everything in flight. The loop body has no branch (so hipcc's
// counted waits survive: tflite_exec.hip tells what happens otherwise); iterations past the end of K load the pixel operand from
// past the buffer (zeros) and add nothing.
#include <hip/hip_runtime.h>

#include <climits>

#include "yh_internal.h"

namespace yh {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DT = 32;        // tile edge (channels and pixels)
constexpr int RS = DT + 4;    // f32 row stride of a wave's partial tile in LDS (16-byte aligned rows, odd multiple of 4 banks)

// KK: kernel extent (1 or 3); D: k-steps per iteration (KK == 3: the nine taps of a 16-channel slice; KK == 1: D slices);
// NR: rings; KS: waves that share a tile's K (1 or 4)
template <int KK, int D, int NR, int KS>
__global__ __launch_bounds__(64 * KS) void conv_direct_f16(const ConvParams p) {
    static_assert((KK == 3 && D == 9) || (KK == 1 && (D == 1 || D == 2 || D == 4 || D == 8)), "steps per iteration");
    static_assert((NR == 2 || NR == 3) && (KS == 1 || KS == 4), "rings / waves per tile");
    __shared__ __attribute__((aligned(16))) float red[KS == 4 ? 4 * DT * RS : 4];   // [wave][pixel][channel]
    const int tid = threadIdx.x, lane = tid & 63, w = KS == 4 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    const int l31 = lane & 31, g = lane >> 5;
    const int tile_id = blockIdx.x;
    const int ch_tile = tile_id % p.n_ch_tiles + p.ch_tile0, m_tile = tile_id / p.n_ch_tiles + p.m_tile0;
    const int PQ = p.P * p.Q;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    // B operand: pixel l31 of the tile - one byte offset per tap, computed once
    const int m = m_tile * DT + l31;
    const bool live = m < p.M;
    const int n = live ? m / PQ : 0, rem = live ? m - n * PQ : 0, op = rem / p.Q, oq = rem - op * p.Q;
    const int iy0 = op * p.stride - p.pad, ix0 = oq * p.stride - p.pad;
    constexpr unsigned kPast = 0x80000000u;   // (an offset past every buffer: the load returns zeros)
    unsigned voff_b[KK * KK];
#pragma unroll
    for (int r = 0; r < KK; ++r)
#pragma unroll
        for (int s_ = 0; s_ < KK; ++s_) {
            const int iy = iy0 + r, ix = ix0 + s_;
            const bool in = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            voff_b[r * KK + s_] = in ? (unsigned)(n * p.x_img_stride + (iy * p.W + ix) * p.C + 8 * g + 16 * w) * 2u : kPast;
        }
    // A operand: channel l31 of the tile (the panel has coutPad rows)
    const unsigned voff_a = (unsigned)((ch_tile * DT + l31) * p.ldw + 8 * g + 16 * w) * 2u;
    const int tapC2 = p.C * 2;   // bytes between consecutive taps of a weight row
    const int nit = (p.C >> 4) / (KS * (KK == 3 ? 1 : D));   // (the launch guarantees the division is exact)
    struct Tile { u32x4 a, b; };
    // step d of iteration it: KK == 3: tap d of slice it; KK == 1: slice it * D + d. (A slice = 16 channels per wave: 32 bytes x KS.)
    auto fetch = [&](int it, int d) {
        Tile t;
        const int cbytes = (KK == 3 ? it : it * D + d) * (32 * KS);
        // (an iteration past the end: the pixel operand comes from past the buffer - zeros; the weight operand is whatever finite
        // weights or zeros lie there, times zero)
        const unsigned vb = it < nit ? voff_b[KK == 3 ? d : 0] : kPast;
        t.a = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff_a, (KK == 3 ? d * tapC2 : 0) + cbytes, 0);
        t.b = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vb, cbytes, 0);
        return t;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    auto mac = [&](const Tile& t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, t.a), __builtin_bit_cast(half8, t.b), acc, 0, 0, 0);
    };
    Tile ring[NR][D];
#pragma unroll
    for (int r = 0; r < NR - 1; ++r)
#pragma unroll
        for (int d = 0; d < D; ++d) ring[r][d] = fetch(r, d);
    for (int it = 0; it < nit; it += NR) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                ring[(r + NR - 1) % NR][d] = fetch(it + r + NR - 1, d);
                mac(ring[r][d]);
            }
    }
    // ---- epilogue. C/D layout of the 32 x 32 MFMA (weights = A): register 4 q + e <-> channel 8 q + 4 g + e, pixel l31
    auto finish = [&](int mo, int ch, f32x4 v) {   // four consecutive channels of one output pixel
        if (mo >= p.M || ch >= p.cout8) return;
        v += *(const f32x4*)(p.bias + ch);   // (bias is padded to coutPad)
        long long yo, ro;
        if (p.y_dense) { yo = (long long)mo * p.ldy + ch; ro = (long long)mo * p.ldres + ch; }
        else {
            const int no = mo / PQ, ro_ = mo - no * PQ;
            yo = no * p.y_img_stride + (long long)ro_ * p.ldy + ch;
            ro = no * p.res_img_stride + (long long)ro_ * p.ldres + ch;
        }
        if (p.res) {
            const half4 r = *(const half4*)(p.res + ro);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *(half4*)(p.y + yo) = o;
    };
    if constexpr (KS == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            finish(m_tile * DT + l31, ch_tile * DT + 8 * q + 4 * g, f32x4{ acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3] });
    } else {
        float* const mine = red + w * (DT * RS) + l31 * RS;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(mine + 8 * q + 4 * g) = f32x4{ acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3] };
        __syncthreads();
        // thread -> pixel tid >> 3, channels 4 (tid & 7) .. + 3; the partial tiles are added in wave order
        const int px = tid >> 3, c4 = (tid & 7) * 4;
        f32x4 v = *(const f32x4*)(red + px * RS + c4);
#pragma unroll
        for (int k = 1; k < 4; ++k) v += *(const f32x4*)(red + k * (DT * RS) + px * RS + c4);
        finish(m_tile * DT + px, ch_tile * DT + c4, v);
    }
}

template <int KK, int D, int NR, int KS>
void launch_form(const ConvParams& p, unsigned tiles, hipStream_t stream) {
    hipLaunchKernelGGL((conv_direct_f16<KK, D, NR, KS>), dim3(tiles), dim3(64 * KS), 0, stream, p);
}

}  // namespace

// Can this convolution run on the direct kernel? (plain 1x1 and 3x3 f16 convolutions: no fp8 output, no fused upsample / second
// source / levels / tail, no tanh channels, input channels a multiple of 64, output rows 8-byte aligned)
bool conv_direct_ok(const ConvParams& p) {
    return p.y && !p.y8 && !p.scale && !p.res_up && !p.nlev && !p.x2 && !p.w2 && p.tanh_from == INT_MAX && p.k_slices <= 1 && p.C % 64 == 0 && ((p.R == 1 && p.S == 1) || (p.R == 3 && p.S == 3)) &&
           p.cout8 % 4 == 0 && p.ldy % 4 == 0 && (!p.res || p.ldres % 4 == 0) && !p.skip_dma;
}

// Which form: K split over four waves where every wave still has a loop to run (3x3: 128 input channels on; 1x1: 512 on), else one
// wave per tile; three rings where a wave has more than two iterations. `form` (study knob, ConvParams::k1steps of a plain launch):
// bit 0 two rings always, bit 1 one wave per tile always.
hipError_t launch_conv_direct(const ConvParams& p, hipStream_t stream) {
    if (!conv_direct_ok(p)) return hipErrorInvalidValue;
    const int n_m_tiles = (p.M + DT - 1) / DT - p.m_tile0;
    if (n_m_tiles < 1 || p.n_ch_tiles < 1) return hipErrorInvalidValue;
    const unsigned tiles = (unsigned)(n_m_tiles * p.n_ch_tiles);
    const int slices = p.C >> 4;   // 16-channel slices of the input (a multiple of 4)
    const bool two = p.k1steps & 1, one = p.k1steps & 2;
    if (p.R == 3) {
        if (slices >= 8 && !one) { if (slices / 4 > 2 && !two) launch_form<3, 9, 3, 4>(p, tiles, stream); else launch_form<3, 9, 2, 4>(p, tiles, stream); }
        else if (!two) launch_form<3, 9, 3, 1>(p, tiles, stream);
        else launch_form<3, 9, 2, 1>(p, tiles, stream);
    } else if (slices >= 32 && !one) {   // 1x1, K >= 512: four waves
        const int per = slices / 4;
        if (per % 8 == 0) { if (per / 8 > 2 && !two) launch_form<1, 8, 3, 4>(p, tiles, stream); else launch_form<1, 8, 2, 4>(p, tiles, stream); }
        else if (per % 4 == 0) launch_form<1, 4, 3, 4>(p, tiles, stream);
        else launch_form<1, 1, 3, 4>(p, tiles, stream);
    } else {   // 1x1, short K (or forced): one wave per tile
        if (slices % 8 == 0) { if (slices / 8 > 2 && !two) launch_form<1, 8, 3, 1>(p, tiles, stream); else launch_form<1, 8, 2, 1>(p, tiles, stream); }
        else launch_form<1, 4, 2, 1>(p, tiles, stream);
    }
    return hipGetLastError();
}

}  // namespace yh
