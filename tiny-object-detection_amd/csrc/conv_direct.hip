// conv_direct_f16 - small convolutions (few output pixels, long K: ResNet layers 3-4 at batch 1-4) with NO LDS staging, no
// barrier in the K loop and no split-K launch pair.
//
// What the LDS-tiled kernels (conv_igemm.hip) cost at batch 1 is not their MFMAs: a layer-3 3x3 convolution (1225 pixels x 256
// channels, K = 2304) has 80 tiles of 64 x 64, so it is split six ways along K, every slice writes a 16 KB f32 slab and a second
// launch sums the slabs - 11 us + 6.5 us for 1.45 GFLOP, on a chip where no launch takes less than 4.4 us. The TFLite executor's
// int8 convolutions had the same shape and went from 13.5 to 6 us per launch by feeding the MFMAs straight from global memory
// (tflite_exec.hip, tfl_conv_i8_direct); this is the f16 form of that kernel.
//
// A workgroup = 32 output channels x 32 output pixels, four waves. Every wave computes the WHOLE tile for a quarter of K - of each
// 64-channel chunk of the input, wave w takes channels 16 w .. 16 w + 15 (the four waves together read one 128-byte line of a row) -
// with v_mfma_f32_32x32x16_f16, whose operand registers are plain 16-byte buffer loads: lane (l31, g) holds K elements 8 g .. 8 g + 7
// of row l31, contiguous both in the weight panel [Co][(r, s, c)] and in an NHWC pixel. The K loop is as bare as the hardware
// allows: the tap of a step is a compile-time constant (KK = 1 or 3: an iteration of a 3x3 launch is the nine taps of one channel
// chunk), so every per-lane address is computed ONCE - one byte offset per tap for the pixel (a tap outside the image, or a row
// past M, gets an offset past the buffer: the load returns zeros, there is no select), one for the weight row - and a step is two
// buffer loads whose scalar offset carries the channel chunk, one MFMA and the scalar adds. The first version computed positions
// and selects per step: 63 instructions per MFMA, 22 us for a launch the tiled kernel does in 7. D steps of loads are in flight
// (a ring of registers indexed at compile time); the loop body has no branch (so hipcc's counted waits survive: tflite_exec.hip
// tells what happens otherwise) and the last iteration is peeled, so nothing is fetched past K. The four partial tiles meet ONCE,
// in LDS (18 KB), are added in wave order and go through the usual epilogue (bias, residual, ReLU) to an 8-byte f16 store per
// lane. Operand traffic is what a 32 x 32 tile costs ((32 + 32) x K x 2 bytes per workgroup, L2 hits: the layer's operands are
// 1-5 MB), there is no slab and no second launch.
// Measured: faster than the tiled launch + reduce only on the six smallest launches of a batch-1 step,
// slower on everything larger, neutral per step - opt-in (yh_tuning.direct). A second version with three rotating rings (36 loads
// in flight per wave, the machine scheduler pinned with sched_barrier so that the counted waits came out as vmcnt(36)) and a
// one-wave form for short K was SLOWER still (a 312-workgroup 3x3: 30 us against 20.5 here and 17 tiled): a wave-load of this
// kernel touches 32 weight rows x 32 bytes - a quarter of each 128-byte line, one DRAM page per row - where the tiled kernel's
// LDS-DMA reads whole lines; more of such requests in flight is not more bandwidth. The int8 form wins because its layers are L2-
// resident and tiny; these are GFLOP-sized.
#include <hip/hip_runtime.h>

#include <climits>

#include "yh_internal.h"

namespace yh {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DT = 32;        // tile edge (channels and pixels)
constexpr int RS = DT + 4;    // f32 row stride of a wave's partial tile in LDS (16-byte aligned rows, odd multiple of 4 banks)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KK, int D>   // KK: kernel extent (1 or 3); D: k-steps per loop iteration (KK == 3: the nine taps; KK == 1: 1, 2, 4 or 8 channel chunks)
__global__ __launch_bounds__(256) void conv_direct_f16(const ConvParams p) {
    static_assert((KK == 3 && D == 9) || (KK == 1 && (D == 1 || D == 2 || D == 4 || D == 8)), "steps per iteration");
    __shared__ __attribute__((aligned(16))) float red[4 * DT * RS];   // [wave][pixel][channel]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    unsigned voff_b[KK * KK];
#pragma unroll
    for (int r = 0; r < KK; ++r)
#pragma unroll
        for (int s_ = 0; s_ < KK; ++s_) {
            const int iy = iy0 + r, ix = ix0 + s_;
            const bool in = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            voff_b[r * KK + s_] = in ? (unsigned)(n * p.x_img_stride + (iy * p.W + ix) * p.C + 8 * g + 16 * w) * 2u : 0x80000000u;   // (past the buffer: zeros)
        }
    // A operand: channel l31 of the tile (the panel has coutPad rows)
    const unsigned voff_a = (unsigned)((ch_tile * DT + l31) * p.ldw + 8 * g + 16 * w) * 2u;
    const int tapC2 = p.C * 2;   // bytes between consecutive taps of a weight row
    struct Tile { u32x4 a, b; };
    // step d of iteration it: KK == 3: tap d, channel chunk it; KK == 1: channel chunk it * D + d
    auto fetch = [&](int it, int d) {
        Tile t;
        const int cbytes = (KK == 3 ? it : it * D + d) * 128;
        t.a = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff_a, (KK == 3 ? d * tapC2 : 0) + cbytes, 0);
        t.b = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff_b[KK == 3 ? d : 0], cbytes, 0);
        return t;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    auto mac = [&](const Tile& t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, t.a), __builtin_bit_cast(half8, t.b), acc, 0, 0, 0);
    };
    const int nit = (p.C >> 6) / (KK == 3 ? 1 : D);   // (the launch guarantees the division is exact)
    // Two rings, ping-pong: iteration it is consumed out of one ring while iteration it + 1 is loaded into the other. (One ring
    // refilled slot by slot - load into ring[d] while its old value feeds the MFMA - made the register allocator rotate the ring with
    // v_mov copies at the top of the loop, and a copy of a register that was just loaded is a wait for everything in flight.)
    Tile ra[D], rb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ra[d] = fetch(0, d);
    int it = 1;
    for (; it + 1 < nit; it += 2) {
#pragma unroll
        for (int d = 0; d < D; ++d) { rb[d] = fetch(it, d); mac(ra[d]); }
#pragma unroll
        for (int d = 0; d < D; ++d) { ra[d] = fetch(it + 1, d); mac(rb[d]); }
    }
    if (it < nit) {   // (one more iteration: an even count)
#pragma unroll
        for (int d = 0; d < D; ++d) { rb[d] = fetch(it, d); mac(ra[d]); }
#pragma unroll
        for (int d = 0; d < D; ++d) mac(rb[d]);
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) mac(ra[d]);
    }
    // C/D layout of the 32 x 32 MFMA (weights = A): register 4 q + e <-> channel 8 q + 4 g + e, pixel l31
    float* const mine = red + w * (DT * RS) + l31 * RS;
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(mine + 8 * q + 4 * g) = f32x4{ acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3] };
    __syncthreads();
    // epilogue: thread -> pixel tid >> 3, channels 4 (tid & 7) .. + 3
    const int px = tid >> 3, c4 = (tid & 7) * 4;
    const int mo = m_tile * DT + px, ch = ch_tile * DT + c4;
    if (mo >= p.M || ch >= p.cout8) return;
    f32x4 v = *(const f32x4*)(red + px * RS + c4);
#pragma unroll
    for (int k = 1; k < 4; ++k) v += *(const f32x4*)(red + k * (DT * RS) + px * RS + c4);
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
}

}  // namespace

// Can this convolution run on the direct kernel? (plain 1x1 and 3x3 f16 convolutions: no fp8 output, no fused upsample / second source / levels /
// tail, no tanh channels, input channels a multiple of 64, output rows 8-byte aligned)
bool conv_direct_ok(const ConvParams& p) {
    return p.y && !p.y8 && !p.scale && !p.res_up && !p.nlev && !p.x2 && !p.w2 && p.tanh_from == INT_MAX && p.k_slices <= 1 && p.C % 64 == 0 && ((p.R == 1 && p.S == 1) || (p.R == 3 && p.S == 3)) &&
           p.cout8 % 4 == 0 && p.ldy % 4 == 0 && (!p.res || p.ldres % 4 == 0) && !p.skip_dma;
}

hipError_t launch_conv_direct(const ConvParams& p, hipStream_t stream) {
    if (!conv_direct_ok(p)) return hipErrorInvalidValue;
    const int n_m_tiles = (p.M + DT - 1) / DT - p.m_tile0;
    if (n_m_tiles < 1 || p.n_ch_tiles < 1) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(n_m_tiles * p.n_ch_tiles)), block(256);
    const int chunks = p.C >> 6;
    if (p.R == 3) hipLaunchKernelGGL((conv_direct_f16<3, 9>), grid, block, 0, stream, p);
    else if (chunks % 8 == 0) hipLaunchKernelGGL((conv_direct_f16<1, 8>), grid, block, 0, stream, p);
    else if (chunks % 4 == 0) hipLaunchKernelGGL((conv_direct_f16<1, 4>), grid, block, 0, stream, p);
    else if (chunks % 2 == 0) hipLaunchKernelGGL((conv_direct_f16<1, 2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_direct_f16<1, 1>), grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace yh
