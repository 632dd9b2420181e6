// bneck.hip — a whole ResNet bottleneck tail as ONE kernel ("cut inter-kernel HBM traffic in layers 1-2").
//
// Inside interpreter.invoke() (/root/reference/src/yolact.rs:163) a bottleneck block is CONV_2D -> CONV_2D -> CONV_2D + ADD
// (+ the next block's first CONV_2D). As separate launches every block writes and re-reads its 64 / 128-channel intermediates
// and its 4x-wide output: 16 tensor-units of HBM traffic per identity block (unit = pixels x planes x 2 B; at batch 64 layer 1
// moves 2.5 GB per block at 0.49 of the HBM peak, 32 % of the step). This kernel runs, for one tile of TM output pixels,
//
//   b  = relu(conv3x3(a; stride, pad 1) + bias_b)                       planes -> planes      (kept in LDS)
//   y  = relu(W_c b + bias_c + x)                                        planes -> 4 planes    (written: the next block's residual)
//   a' = relu(W_a' y + bias_a')                                          4 planes -> planes    (written: the next block's conv_b input)
//
// so a block reads a (1 unit, halo through L2) and x (4) and writes y (4) and a' (1): 10 units instead of 16, and three
// launches become one. The 1x1 convolutions need no halo, so nothing is recomputed; the 3x3 loads its taps through the
// normal implicit-GEMM loader.
//
// Bit-transparent by construction: every output element accumulates the same v_mfma_f32_32x32x16_f16 products in the same
// order as conv_igemm_f16's streaming tiles (k = 64-channel chunk outer, taps inner, four 16-deep slices per step), the
// epilogues apply the same f32 operations in the same order (acc + bias, + residual, max 0, round to f16), and the
// intermediates are rounded to f16 exactly where the separate launches store them. tests/test_gpu_bneck.py compares the
// fused engine with the unfused one bit for bit.
//
// LDS (one array; PL = 64, TM = 256: 80 KB -> two workgroups per CU):
//   [0, A)        phase 1: weight tile [PL][128 B] + activation tile [TM][128 B] of the current k-step (single stage, as the
//                 streaming tiles); afterwards Wt2 (the expand conv's weight k-tiles of the current 64-channel chunk)
//                 and Bt = b as PL/64 activation k-tiles [TM][128 B]
//   [A, A + S)    stage [TM][128 B]: one 64-channel chunk of a row - first the residual (row loads), then in place the rounded y
//                 chunk, which is both stored from here with whole-row 16-byte stores and read back as the B operand of a'
//   [.., + W3)    Wt3: W_a' k-tile [PL][128 B] of the current chunk
// Every tile row is 128 bytes with logical 16-byte chunk c of row r at physical chunk c ^ ((r >> 1) & 7): the LDS-DMA image
// is lane-linear, so the permutation is on each lane's SOURCE address (conv_igemm.hip), and ds_read_b128 fragment reads are
// conflict free.
#include "yh_internal.h"

namespace yh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
typedef __attribute__((address_space(3))) char lds_char;

template <int A, int B>
struct bmax { static constexpr int v = A > B ? A : B; };

// vmcnt(N) (all but the N youngest vector-memory operations of this wave have completed), lgkmcnt(0), workgroup barrier.
// Four bias values bias[base8 + 4 lh .. + 3] (base8 wave-uniform, a multiple of 8) through the SCALAR cache: the constant address
// space makes the uniform-index loads s_load_dwordx8 (lgkmcnt), and the lane picks its half. A vector load here would be one more
// vmcnt entry YOUNGER than the residual rows and weight tiles requested ahead, and the wait in front of its use (vmcnt(0) - seen
// in the ISA after every bias load of the first version) made each chunk wait for the next chunk's residual rows to arrive
// from HBM: the prefetch undone.
typedef const __attribute__((address_space(4))) float cfloat_t;
__device__ __forceinline__ f32x4 bias_quad(const float* bias, int base8, int lh) {
    cfloat_t* const b = (cfloat_t*)(unsigned long long)bias;
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = b[base8 + e]; hi[e] = b[base8 + 4 + e]; }
    return lh ? hi : lo;
}

template <int N>
__device__ __forceinline__ void bar_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
}  // namespace

// WCH x WM: wave grid of the PL-channel GEMMs (conv_b, a'); W2C x W2M: of the 64-channel chunk GEMM (the expand conv).
// KT2 > 0: the stage's FIRST block - no identity shortcut; the expand conv continues its K over KT2 64-channel tiles of a second
// tensor x2 (the block input, read at stride2): y = relu([W_c | W_d] [b ; x2] + (bias_c + bias_d)), the two-source form of
// conv_igemm_f16 (ConvParams::x2) with b coming out of LDS. The x2 tile is fetched once per tile and serves every chunk.
template <int PL, int TM, int WCH, int WM, int W2C, int W2M, bool NEXT, int KT2 = 0>
__global__ __launch_bounds__(256, 2) void bneck_chain_f16(const BneckParams p) {
    static_assert(PL % 64 == 0 && TM % 32 == 0 && WCH * WM == 4 && W2C * W2M == 4, "four waves");
    constexpr int KT = PL / 64;                     // 64-channel k-tiles of a PL-channel tensor
    constexpr int WTC = PL / WCH, WTM = TM / WM, TC = WTC / 32, TMT = WTM / 32;          // PL-channel GEMMs
    constexpr int W2TC = 64 / W2C, W2TM = TM / W2M, TC2 = W2TC / 32, TMT2 = W2TM / 32;   // chunk GEMM
    static_assert(WTC % 32 == 0 && WTM % 32 == 0 && W2TC % 32 == 0 && W2TM % 32 == 0, "wave tiles are multiples of the 32 x 32 MFMA tile");
    constexpr int XL = TM / 32, WL = PL / 32;       // LDS-DMA passes (32 rows each) of an activation / PL-row weight tile
    constexpr int NOC = (4 * PL) / 64;              // 64-channel chunks of the block output
    constexpr int P1_BYTES = (PL + TM) * 128;       // phase 1: one k-step's weight + activation tile
    constexpr int WT2_BYTES = (KT + KT2) * 64 * 128, BT_BYTES = KT * TM * 128;
    constexpr int A_BYTES = bmax<P1_BYTES, WT2_BYTES + BT_BYTES>::v;
    constexpr int ST_BYTES = TM * 128, WT3_BYTES = NEXT ? PL * 128 : 0, X2_BYTES = KT2 * TM * 128;
    // (phase 1 double-buffers its k-step tiles over regions A + stage + Wt3, which are idle until phase 2: X2t stays clear of them)
    constexpr int LDS_BYTES = bmax<A_BYTES + ST_BYTES + WT3_BYTES, 2 * P1_BYTES>::v + X2_BYTES;
    static_assert(A_BYTES + ST_BYTES + WT3_BYTES >= 2 * P1_BYTES || X2_BYTES == 0, "the second source's tile must not overlap phase 1's second buffer");
    constexpr int LDW3 = PL + 64 * KT2;             // row stride of the expand conv's panel ([W_c | W_d] in the two-source form)
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    lds_char* const lds3 = (lds_char*)lds;
    char* const wt2 = lds;                          // after phase 1
    char* const bt = lds + WT2_BYTES;
    char* const stage = lds + A_BYTES;
    char* const wt3 = lds + A_BYTES + ST_BYTES;
    char* const x2t = lds + A_BYTES + ST_BYTES + WT3_BYTES;

    // (one workgroup per tile; the loop form is kept from the persistent-grid experiment of round 3, which measured slower: DESIGN.md section 4)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int ntiles = (p.M + TM - 1) / TM;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pc = tid & 7, rb = tid >> 3;          // DMA / row-pass role: physical chunk pc of rows rb + 32 d
    const int lc = pc ^ ((rb >> 1) & 7);            // ... holds logical chunk lc
    const int l31 = lane & 31, lh = lane >> 5, swz = (l31 >> 1) & 7;
    const int PQ = p.P * p.Q;
    const int C4 = 4 * PL;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(NEXT ? p.w1n : p.w3), 0, (int)(NEXT ? p.w1n_bytes : 0u), 0x00020000);

    for (int tile = bid; tile < ntiles; tile += nwg) {
    const int m0 = tile * TM;
    // ---- residual rows of chunk oc -> registers (16 bytes per lane and pass: rows rb + 32 d, logical chunk lc)
    half8 resv[KT2 ? 1 : XL];
    auto load_res = [&](int oc) {
        if (KT2) return;
#pragma unroll
        for (int d = 0; d < XL; ++d) {
            const int m = m0 + rb + 32 * d;
            resv[d] = *(const half8*)(p.res + (long long)(m < p.M ? m : 0) * C4 + oc * 64 + lc * 8);
        }
    };
    load_res(0);   // (lands under phase 1)
    if (KT2) {   // the second source's rows of this tile -> X2t (LDS-DMA; waited for by phase 1's first barrier)
        const __amdgpu_buffer_rsrc_t x2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, (int)p.x2_bytes, 0x00020000);
        lds_char* const dstx = lds3 + (A_BYTES + ST_BYTES + WT3_BYTES) + wave * 1024;
#pragma unroll
        for (int d = 0; d < XL; ++d) {
            const int mm = m0 + rb + 32 * d, m = mm < p.M ? mm : 0;
            const int n = m / PQ, rem = m - n * PQ, op = rem / p.Q, oq = rem - op * p.Q;
            const unsigned off = (unsigned)((int)(n * p.x2_img_stride) + (op * p.W2 + oq) * p.stride2 * (64 * KT2) + lc * 8) * 2u;
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rsrc, dstx + kt * (TM * 128) + d * 4096, 16, (int)(off + kt * 128u), 0, 0, 0);
        }
    }

    // ---- phase 1: b = relu(conv3x3(a) + bias_b), streaming-tile loop (one LDS stage; the co-resident workgroup overlaps)
    int xbase[XL], xih[XL], xiw[XL];
#pragma unroll
    for (int d = 0; d < XL; ++d) {
        const int m = m0 + rb + 32 * d;
        if (m < p.M) {
            const int n = m / PQ, rem = m - n * PQ, op = rem / p.Q, oq = rem - op * p.Q;
            xih[d] = op * p.stride - 1;
            xiw[d] = oq * p.stride - 1;
            xbase[d] = (int)(n * p.a_img_stride) + (xih[d] * p.W + xiw[d]) * PL + lc * 8;
        } else { xih[d] = -(1 << 24); xiw[d] = 0; xbase[d] = 0; }
    }
    const int a1_row = ((wave / WM) * WTC + l31) * 128, b1_row = ((wave % WM) * WTM + l31) * 128;
    f32x16 acc[TC][TMT];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TMT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    {
        const unsigned wbase = (unsigned)((rb * (9 * PL) + lc * 8) * 2);
        // Two LDS stages: the DMA of step t + 1 is issued before the MFMAs of step t and lands underneath them; one barrier per
        // step (vmcnt(0) + s_barrier: tile t + 1 has landed, every wave is done reading tile t). Same k order, same bits.
        constexpr int NS = 9 * KT;
        auto issue = [&](int t, int buf) {
            const int kc = (t / 9) * 64, tap = t - (t / 9) * 9, r = tap / 3, s_ = tap - r * 3;
            lds_char* const dstw = lds3 + buf * P1_BYTES + wave * 1024;
            const int wk = (r * 3 + s_) * PL + kc;      // K index of this step in the [(r, s, c)] panel
#pragma unroll
            for (int d = 0; d < WL; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, dstw + d * 4096, 16, (int)(wbase + (unsigned)((32 * d) * (9 * PL) + wk) * 2u), 0, 0, 0);
            const int toff = (r * p.W + s_) * PL + kc;
#pragma unroll
            for (int d = 0; d < XL; ++d) {
                const bool ok = (unsigned)(xih[d] + r) < (unsigned)p.H && (unsigned)(xiw[d] + s_) < (unsigned)p.W;
                const unsigned voff = ok ? (unsigned)(xbase[d] + toff) * 2u : p.a_zero_off;   // padded taps read the allocation's zero block
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dstw + PL * 128 + d * 4096, 16, (int)voff, 0, 0, 0);
            }
        };
        issue(0, 0);
        __syncthreads();
        for (int t = 0; t < NS; ++t) {
            const char* const base = lds + (t & 1) * P1_BYTES;
            if (t + 1 < NS) issue(t + 1, (t + 1) & 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = ((2 * kk + lh) ^ swz) << 4;
                half8 fa[TC], fb[TMT];
#pragma unroll
                for (int i = 0; i < TC; ++i) fa[i] = *(const half8*)(base + a1_row + i * 4096 + co);
#pragma unroll
                for (int j = 0; j < TMT; ++j) fb[j] = *(const half8*)(base + PL * 128 + b1_row + j * 4096 + co);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    __syncthreads();   // region A is reused
    // the expand conv's weight k-tiles of chunk 0 (rows 0 .. 63 of W_c) travel while b is rounded into LDS
    auto dma_wt2 = [&](int oc) {
        lds_char* const dstw = lds3 + wave * 1024;
#pragma unroll
        for (int kt = 0; kt < KT + KT2; ++kt)
#pragma unroll
            for (int d = 0; d < 2; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3_rsrc, dstw + kt * 8192 + d * 4096, 16, (int)((unsigned)(((oc * 64 + rb + 32 * d) * LDW3 + kt * 64 + lc * 8) * 2)), 0, 0, 0);
    };
    auto dma_wt3 = [&](int oc) {
        if (!NEXT) return;
        lds_char* const dstw = lds3 + (A_BYTES + ST_BYTES) + wave * 1024;
#pragma unroll
        for (int d = 0; d < WL; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, dstw + d * 4096, 16, (int)((unsigned)(((rb + 32 * d) * C4 + oc * 64 + lc * 8) * 2)), 0, 0, 0);
    };
    constexpr int N3 = NEXT ? WL : 0;   // LDS-DMA instructions per wave of dma_wt3 (dma_wt2: 2 KT)
    dma_wt2(0);
    {   // b -> Bt: bias, ReLU, round; lane holds pixel (wm, j, l31) and channels (wc, i, 8 g + 4 lh + e)
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (wave / WM) * WTC + i * 32 + 8 * g + 4 * lh;
                const f32x4 b4 = bias_quad(p.bias2, (wave / WM) * WTC + i * 32 + 8 * g, lh);
#pragma unroll
                for (int j = 0; j < TMT; ++j) {
                    const int m = (wave % WM) * WTM + j * 32 + l31;
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)fmaxf(acc[i][j][4 * g + e] + b4[e], 0.0f);
                    *(half4*)(bt + (ch >> 6) * (TM * 128) + m * 128 + ((((ch & 63) >> 3) ^ ((m >> 1) & 7)) << 4) + (ch & 4) * 2) = o;
                }
            }
    }
    // ---- phases 2 + 3, one 64-channel chunk of the block output at a time
    f32x16 acc3[NEXT ? TC : 1][NEXT ? TMT : 1];
    if (NEXT) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TMT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc3[i][j][e] = 0.0f;
    }
    const int a2_row = ((wave / W2M) * W2TC + l31) * 128, b2_row = ((wave % W2M) * W2TM + l31) * 128;
#pragma unroll 1
    for (int oc = 0; oc < NOC; ++oc) {
        // S0: the residual rows of this chunk -> stage (they were requested one chunk ago); a' weights of this chunk on their way
        if (!KT2) {
#pragma unroll
            for (int d = 0; d < XL; ++d) *(half8*)(stage + (rb + 32 * d) * 128 + pc * 16) = resv[d];
        }
        dma_wt3(oc);
        bar_vm<N3>();   // S1: Wt2(oc) has landed (issued before Wt3), the residual and (chunk 0) Bt are visible
        // S3 (early): request the next chunk's residual rows - resv was consumed at S0 - in flight until the next S0
        if (oc + 1 < NOC) load_res(oc + 1);
        // S2 + S4, one 32-channel half of the chunk at a time (half the accumulator registers: the whole chunk at once put the
        // NEXT form at 256 VGPRs with spills, whose scratch reloads serialise against the hand-placed DMA):
        //   acc2[32 ch][TM] = W_c[oc*64 + 32 i .. + 31][:] x [b ; x2], then y = relu(acc2 + bias_c (+ x)), rounded, in place over
        //   the residual in stage (one lane per element). Every element sums its products in the same order as before.
#pragma unroll 1
        for (int i = 0; i < TC2; ++i) {
            f32x16 acc2[TMT2];
#pragma unroll
            for (int j = 0; j < TMT2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[j][e] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < KT + KT2; ++kt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int co = ((2 * kk + lh) ^ swz) << 4;
                    half8 fb[TMT2];
                    const char* const bsrc = kt < KT ? bt + kt * (TM * 128) : x2t + (kt - KT) * (TM * 128);   // b, then the second source
                    const half8 fa = *(const half8*)(wt2 + kt * 8192 + a2_row + i * 4096 + co);
#pragma unroll
                    for (int j = 0; j < TMT2; ++j) fb[j] = *(const half8*)(bsrc + b2_row + j * 4096 + co);
#pragma unroll
                    for (int j = 0; j < TMT2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], acc2[j], 0, 0, 0);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (wave / W2M) * W2TC + i * 32 + 8 * g + 4 * lh;   // channel within the chunk
                const f32x4 b4 = bias_quad(p.bias3, oc * 64 + (wave / W2M) * W2TC + i * 32 + 8 * g, lh);
#pragma unroll
                for (int j = 0; j < TMT2; ++j) {
                    const int m = (wave % W2M) * W2TM + j * 32 + l31;
                    half4* const q = (half4*)(stage + m * 128 + (((ch >> 3) ^ ((m >> 1) & 7)) << 4) + (ch & 4) * 2);
                    half4 r4;
                    if (!KT2) r4 = *q;
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc2[j][4 * g + e] + b4[e];
                        if (!KT2) v = v + (float)r4[e];
                        o[e] = (half_t)fmaxf(v, 0.0f);
                    }
                    *q = o;
                }
            }
        }
        // S5: the y chunk is complete; Wt3(oc) has landed (only the residual loads of S3 may still be in flight)
        if (oc + 1 < NOC && !KT2) bar_vm<XL>(); else bar_vm<0>();
        // S6: whole-row stores of the y chunk; the next chunk's expand weights; a' += W_a'[:, chunk] x y chunk
#pragma unroll
        for (int d = 0; d < XL; ++d) {
            const int m = m0 + rb + 32 * d;
            if (m < p.M) *(half8*)(p.y + (long long)m * C4 + oc * 64 + lc * 8) = *(const half8*)(stage + (rb + 32 * d) * 128 + pc * 16);
        }
        if (oc + 1 < NOC) dma_wt2(oc + 1);
        if (NEXT) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = ((2 * kk + lh) ^ swz) << 4;
                half8 fa[TC], fb[TMT];
#pragma unroll
                for (int i = 0; i < TC; ++i) fa[i] = *(const half8*)(wt3 + a1_row + i * 4096 + co);
#pragma unroll
                for (int j = 0; j < TMT; ++j) fb[j] = *(const half8*)(stage + b1_row + j * 4096 + co);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc3[i][j], 0, 0, 0);
            }
        }
        // S7: everyone is done with stage, Wt2 is refilled behind this barrier's successor S1 (counted), Wt3 is free
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (NEXT) {
    // ---- a' = relu(acc3 + bias_a'), rounded, through Bt (dead: [KT][TM][128 B], the layout of a PL-channel row set), whole-row stores
    {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (wave / WM) * WTC + i * 32 + 8 * g + 4 * lh;
                const f32x4 b4 = bias_quad(p.bias1n, (wave / WM) * WTC + i * 32 + 8 * g, lh);
#pragma unroll
                for (int j = 0; j < TMT; ++j) {
                    const int m = (wave % WM) * WTM + j * 32 + l31;
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)fmaxf(acc3[i][j][4 * g + e] + b4[e], 0.0f);
                    *(half4*)(bt + (ch >> 6) * (TM * 128) + m * 128 + ((((ch & 63) >> 3) ^ ((m >> 1) & 7)) << 4) + (ch & 4) * 2) = o;
                }
            }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int d = 0; d < XL; ++d) {
                const int m = m0 + rb + 32 * d;
                if (m < p.M) *(half8*)(p.a_next + (long long)m * PL + kt * 64 + lc * 8) = *(const half8*)(bt + kt * (TM * 128) + (rb + 32 * d) * 128 + pc * 16);
            }
    }
    }
    __syncthreads();   // the next tile's phase 1 reuses region A
    }   // persistent tile loop
}


// ------------------------------------------------------------------------------------------------------------------------
// bneck_xn_f16 - the same idea for a 256-plane block (layer 3) WITHOUT its 3x3 conv: b is a tensor in memory (the 3x3 conv's
// 1.2 MB of weights do not fit a chain, and in fp8 precision it is an fp8 launch of its own), and this launch computes
//
//   y  = relu(W_c b + bias_c + x)        256 -> 1024   (written: the next block's residual)
//   a' = relu(W_a' y + bias_a')          1024 -> 256   (written: the next block's conv_b input; in fp8 precision also / only as E4M3)
//
// for one tile of 64 pixels per workgroup: two launches become one and y is not read back. What the workgroup moves most is
// WEIGHTS - both panels, 1 MB, stream through LDS once per tile (64 KB per 64-channel chunk of y). With four waves doing
// everything that cost ISSUE time first: an LDS-DMA instruction moves 1 KB and held its wave for 110-150 clocks (in-kernel
// s_memtime stamps: 2 000-2 700 clocks per step for a wave's 18), during which that wave issues nothing else. So the workgroup is
// EIGHT waves with two jobs (52 -> 40 us per launch; what bounds it now is the L2 -> LDS path itself, ~25-30 B/clock/CU):
//   waves 0-3 (compute)  GEMM 2: acc2[32 ch x 32 px] = W_c tile x b; + bias + residual, ReLU, f16, in place into the stage |M|
//                        their half of GEMM 3                                                                               |E|
//   waves 4-7 (loaders)  step oc + 1's residual rows and W_c tile by LDS-DMA |M| the y chunk's whole-row stores out of the stage,
//                        step oc + 1's W_a' tile by LDS-DMA, their half of GEMM 3, counted wait for what the next step needs |E|
// (|M|, |E|: the step's two workgroup barriers; one wave of each kind per SIMD; GEMM 3: acc3[64 ch x 32 px] += W_a' tile x the
// staged y chunk on all eight waves). The b tile is held in the compute waves'
// REGISTERS as the expand conv's B fragments (64 VGPRs, loaded once), which leaves the LDS to two stages of every stream
// (2 x 32 KB W_c rows [64][256], 2 x 32 KB W_a' columns [256][64], 2 x 8 KB stage) plus the biases: 149 KB, one workgroup per CU.
// Inside the loop no vector-memory load returns to registers and the kernel has ONE __shared__ object: the compiler's own
// s_waitcnt in front of a register load's use, or in front of a ds_read it cannot tell apart from an LDS-DMA target (separate
// __shared__ arrays give it alias scopes to try), is vmcnt(0) - it would drain the next step's DMAs every step (seen in the ISA).
//
// Same MFMA products in the same order and the same f32 epilogue operations as the two separate launches: bit-identical to them
// (tests/test_gpu_bneck.py).
__global__ __launch_bounds__(512, 1) void bneck_xn_f16(const BneckParams p) {
    constexpr int PL = 256, TM = 64, KT = 4, C4 = 1024, NOC = 16;
    constexpr int W2_BYTES = KT * 64 * 128, W3_BYTES = PL * 128, ST_BYTES = TM * 128;   // 32 KB, 32 KB, 8 KB
    constexpr int ST_OFF = 2 * W2_BYTES + 2 * W3_BYTES, BIAS_OFF = ST_OFF + 2 * ST_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[BIAS_OFF + (C4 + PL) * 4];
    float* const bias_c = (float*)(lds + BIAS_OFF);
    float* const bias_n = bias_c + C4;
    lds_char* const lds3 = (lds_char*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= 4;
    const int wave = wv & 3, lt = tid & 255;
    const int pc = lt & 7, rb = lt >> 3, lc = pc ^ ((rb >> 1) & 7);     // row-pass role: physical chunk pc of rows rb + 32 d holds logical chunk lc
    const int l31 = lane & 31, lh = lane >> 5, swz = (l31 >> 1) & 7;
    const int wc = wave >> 1, wm = wave & 1;                            // compute waves: 2 x 2 (channel half x pixel half) in both GEMMs
    const int m0 = blockIdx.x * TM;

    if (tid < C4 / 4) *(f32x4*)(bias_c + tid * 4) = *(const f32x4*)(p.bias3 + tid * 4);
    else if (tid < C4 / 4 + PL / 4) *(f32x4*)(bias_n + (tid - C4 / 4) * 4) = *(const f32x4*)(p.bias1n + (tid - C4 / 4) * 4);

    // GEMM 3 on all eight waves: wave wv owns pixels 32 (wv & 1) .. + 31 and channels 64 (wv >> 1) .. + 63 of a'
    const int a3_row = ((wv >> 1) * 64 + l31) * 128, b3_row = ((wv & 1) * 32 + l31) * 128;
    f32x16 acc3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[i][e] = 0.0f;
    auto gemm3 = [&](const char* wt3, const char* stage) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int co = ((2 * kk + lh) ^ swz) << 4;
            const half8 fb = *(const half8*)(stage + b3_row + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half8 fa = *(const half8*)(wt3 + a3_row + i * 4096 + co);
                acc3[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc3[i], 0, 0, 0);
            }
        }
    };
    auto wg_barrier = [&]() {   // lgkmcnt(0) + s_barrier
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (loader) {
        const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)p.res_bytes, 0x00020000);
        unsigned roff[2];   // this thread's two residual rows (rows past M read row 0: their results are not stored)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int m = m0 + rb + 32 * d;
            roff[d] = (unsigned)(((m < p.M ? m : 0) * C4 + lc * 8) * 2);
        }
        auto dma_front = [&](int oc) {   // what GEMM 2 of step oc needs: the residual rows (into the stage) and the W_c tile
            lds_char* const ds = lds3 + ST_OFF + (oc & 1) * ST_BYTES + wave * 1024;
#pragma unroll
            for (int d = 0; d < 2; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, ds + d * 4096, 16, (int)(roff[d] + (unsigned)(oc * 128)), 0, 0, 0);
            lds_char* const d2 = lds3 + (oc & 1) * W2_BYTES + wave * 1024;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w3_rsrc, d2 + kt * 8192 + d * 4096, 16, (int)((unsigned)(((oc * 64 + rb + 32 * d) * PL + kt * 64 + lc * 8) * 2)), 0, 0, 0);
        };
        auto dma_back = [&](int oc) {    // what GEMM 3 of step oc needs: the W_a' tile
            lds_char* const d3 = lds3 + 2 * W2_BYTES + (oc & 1) * W3_BYTES + wave * 1024;
#pragma unroll
            for (int d = 0; d < 8; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, d3 + d * 4096, 16, (int)((unsigned)(((rb + 32 * d) * C4 + oc * 64 + lc * 8) * 2)), 0, 0, 0);
        };
        dma_front(0);
        dma_back(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (the W_a' tile may land under step 0's GEMM 2: the loop's first wait covers it)
        wg_barrier();   // |E(-1)|: step 0's residual rows, its W_c tile and the biases are in LDS
#pragma unroll 1
        for (int oc = 0; oc < NOC; ++oc) {
            const char* const stage = lds + ST_OFF + (oc & 1) * ST_BYTES;
            // Counted waits (vmcnt completes in order): each stream is waited for half a step after it was requested.
            if (oc + 1 < NOC) {
                dma_front(oc + 1);   // 10 instructions (those buffers were released by |E(oc - 1)|)
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // all but these: this step's W_a' tile has landed
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();   // |M(oc)|: the y chunk is complete in the stage, GEMM 3 may start
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int m = m0 + rb + 32 * d;
                if (m < p.M) *(half8*)(p.y + (long long)m * C4 + oc * 64 + lc * 8) = *(const half8*)(stage + (rb + 32 * d) * 128 + pc * 16);
            }
            if (oc + 1 < NOC) dma_back(oc + 1);    // 8 instructions
            gemm3(lds + 2 * W2_BYTES + (oc & 1) * W3_BYTES, stage);
            if (oc + 1 < NOC) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // all but those 8: the stores are out, step oc + 1's residual rows and W_c tile have landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();   // |E(oc)|
        }
    } else {
        // b fragments of this wave's 32 pixels: k-tile kt, 16-deep slice kk -> channels kt * 64 + (2 kk + lh) * 8 .. + 7 of row wm * 32 + l31
        half8 bf[KT][4];
        {
            const int mm = m0 + wm * 32 + l31;
            const half_t* brow = p.a + (long long)(mm < p.M ? mm : 0) * PL + lh * 8;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bf[kt][kk] = *(const half8*)(brow + kt * 64 + kk * 16);
        }
        // (used here so that the compiler's wait for them stands in front of the loop, not inside it)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" :: "v"(bf[kt][kk]));
        const int a2_row = (wc * 32 + l31) * 128;
        wg_barrier();   // |E(-1)|
#pragma unroll 1
        for (int oc = 0; oc < NOC; ++oc) {
            const char* const wt2 = lds + (oc & 1) * W2_BYTES;
            const char* const wt3 = lds + 2 * W2_BYTES + (oc & 1) * W3_BYTES;
            char* const stage = lds + ST_OFF + (oc & 1) * ST_BYTES;
            // GEMM 2: 32 channels x 32 pixels per wave, K = 256 out of the b fragments
            f32x16 acc2;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const half8 fa = *(const half8*)(wt2 + kt * 8192 + a2_row + (((2 * kk + lh) ^ swz) << 4));
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, bf[kt][kk], acc2, 0, 0, 0);
                }
            {   // all eight LDS reads of the epilogue first (issued under the MFMAs' tail), then the arithmetic, then the writes
                const int m = wm * 32 + l31;   // pixel within the tile; channels wc * 32 + 8 g + 4 lh .. + 3 of the chunk
                f32x4 b4[4];
                half4 r4[4];
                half4* q[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wc * 32 + 8 * g + 4 * lh;
                    b4[g] = *(const f32x4*)(bias_c + oc * 64 + ch);
                    q[g] = (half4*)(stage + m * 128 + (((ch >> 3) ^ ((m >> 1) & 7)) << 4) + (ch & 4) * 2);
                    r4[g] = *q[g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc2[4 * g + e] + b4[g][e];
                        v = v + (float)r4[g][e];
                        o[e] = (half_t)fmaxf(v, 0.0f);
                    }
                    *q[g] = o;
                }
            }
            wg_barrier();   // |M(oc)|
            gemm3(wt3, stage);
            wg_barrier();   // |E(oc)|
        }
    }
    // a' = relu(acc3 + bias_a'), rounded, into the (dead) first weight stage as [4 k-tiles][64 rows][128 B]: all eight waves
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = (wv >> 1) * 64 + i * 32 + 8 * g + 4 * lh, m = (wv & 1) * 32 + l31;
            const f32x4 b4 = *(const f32x4*)(bias_n + ch);
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)fmaxf(acc3[i][4 * g + e] + b4[e], 0.0f);
            *(half4*)(lds + (ch >> 6) * (TM * 128) + m * 128 + ((((ch & 63) >> 3) ^ ((m >> 1) & 7)) << 4) + (ch & 4) * 2) = o;
        }
    wg_barrier();
    // whole-row stores of a': k-tiles 2 (tid >> 8) and + 1
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int kt = 2 * (tid >> 8) + k2;
        // (fp8: the channels' reciprocal scales, once per k-tile - read inside the row loop they were re-loaded behind every store)
        f32x4 inv0 = { 0.0f, 0.0f, 0.0f, 0.0f }, inv1 = inv0;
        if (p.a_next8) { inv0 = *(const f32x4*)(p.a_next8_inv + kt * 64 + lc * 8); inv1 = *(const f32x4*)(p.a_next8_inv + kt * 64 + lc * 8 + 4); }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int m = m0 + rb + 32 * d;
            if (m < p.M) {
                const half8 o = *(const half8*)(lds + kt * (TM * 128) + (rb + 32 * d) * 128 + pc * 16);
                if (p.a_next) *(half8*)(p.a_next + (long long)m * PL + kt * 64 + lc * 8) = o;
                if (p.a_next8) {   // fp8 precision: a' feeds an fp8 convolution (quantised from the f16-rounded value, as conv_igemm's epilogue)
                    const unsigned lo = e4m3_pack4((float)o[0] * inv0[0], (float)o[1] * inv0[1], (float)o[2] * inv0[2], (float)o[3] * inv0[3]);
                    const unsigned hi = e4m3_pack4((float)o[4] * inv1[0], (float)o[5] * inv1[1], (float)o[6] * inv1[2], (float)o[7] * inv1[3]);
                    *(uint2*)(p.a_next8 + (long long)m * PL + kt * 64 + lc * 8) = make_uint2(lo, hi);
                }
            }
        }
    }
}

const char* bneck_symbol(int planes, int tm, bool next, bool dual) {
    if (planes == 256) return "bneck_xn_f16";
    if (dual) return "bneck_chain_f16<64,128,next,dual>";
    if (planes == 64 && tm == 128) return next ? "bneck_chain_f16<64,128,next>" : "bneck_chain_f16<64,128>";
    if (planes == 64) return tm == 256 ? (next ? "bneck_chain_f16<64,256,next>" : "bneck_chain_f16<64,256>") : (next ? "bneck_chain_f16<64,64,next>" : "bneck_chain_f16<64,64>");
    return tm == 128 ? (next ? "bneck_chain_f16<128,128,next>" : "bneck_chain_f16<128,128>") : (next ? "bneck_chain_f16<128,64,next>" : "bneck_chain_f16<128,64>");
}

// planes in {64, 128}; tm: 256 / 64 for 64 planes, 128 / 64 for 128 planes (the small tiles for launches that would leave CUs idle)
hipError_t launch_bneck(const BneckParams& p, int planes, int tm, hipStream_t stream) {
    if (p.M < 1 || (p.stride != 1 && p.stride != 2)) return hipErrorInvalidValue;
    const int ntiles = (p.M + tm - 1) / tm;
    const dim3 grid((unsigned)ntiles);
    const bool next = p.a_next != nullptr;
    if (p.no_b) {   // expand conv + residual + next reduce conv of a 256-plane block; b is a tensor (p.a)
        if (planes != 256 || tm != 64 || p.x2 || !p.res || !p.w1n || (!p.a_next && !p.a_next8)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(bneck_xn_f16, grid, dim3(512), 0, stream, p);
        return hipGetLastError();
    }
    if (p.x2) {   // the stage's first block (two-source expand conv): 64 planes, 64-channel second source, 128-pixel tiles
        if (planes != 64 || tm != 128 || p.C2 != 64 || !next || p.res) return hipErrorInvalidValue;
        hipLaunchKernelGGL((bneck_chain_f16<64, 128, 1, 4, 1, 4, true, 1>), grid, dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (planes == 64 && tm == 256) {
        if (next) hipLaunchKernelGGL((bneck_chain_f16<64, 256, 1, 4, 1, 4, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bneck_chain_f16<64, 256, 1, 4, 1, 4, false>), grid, dim3(256), 0, stream, p);
    } else if (planes == 64 && tm == 128) {
        if (next) hipLaunchKernelGGL((bneck_chain_f16<64, 128, 1, 4, 1, 4, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bneck_chain_f16<64, 128, 1, 4, 1, 4, false>), grid, dim3(256), 0, stream, p);
    } else if (planes == 64 && tm == 64) {
        if (next) hipLaunchKernelGGL((bneck_chain_f16<64, 64, 2, 2, 2, 2, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bneck_chain_f16<64, 64, 2, 2, 2, 2, false>), grid, dim3(256), 0, stream, p);
    } else if (planes == 128 && tm == 128) {
        if (next) hipLaunchKernelGGL((bneck_chain_f16<128, 128, 2, 2, 1, 4, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bneck_chain_f16<128, 128, 2, 2, 1, 4, false>), grid, dim3(256), 0, stream, p);
    } else if (planes == 128 && tm == 64) {
        if (next) hipLaunchKernelGGL((bneck_chain_f16<128, 64, 2, 2, 2, 2, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bneck_chain_f16<128, 64, 2, 2, 2, 2, false>), grid, dim3(256), 0, stream, p);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace yh
