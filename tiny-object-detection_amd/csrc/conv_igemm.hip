// conv_igemm.hip — NHWC f16 convolution as an implicit GEMM on CDNA4 MFMA (gfx950).
//
// Replaces, for the YOLACT path, the CONV_2D / ADD / PAD / RELU / TANH ops that the reference
// executes inside interpreter.invoke() (/root/reference/src/yolact.rs:163; op histogram
// data/FRC_model_edgetpu.log:7-19): bias, residual add, ReLU and tanh are fused in the epilogue,
// zero padding is folded into the loader (no PAD op, no padded copies).
//
//   D[ch][m] = sum_k Wt[ch][k] * X[m][k]     (weights are the MFMA A operand, activations B)
//
// Work decomposition: one workgroup of 4 or 8 waves (WCH x WM) owns a TCH x TM output tile; each wave
// owns (TCH/WCH) x (TM/WM) of it as 32x32 (v_mfma_f32_32x32x16_f16) or 16x16 (v_mfma_f32_16x16x32_f16)
// MFMA tiles with f32 accumulators. K advances 64 at a time (one 128-byte row segment per tile row):
// global -> LDS directly by LDS-DMA (buffer_load_dwordx4 ... lds: 16 B per lane, 8 lanes per 128-B line, no VGPR
// staging, no ds_write), double buffered, one barrier per step, the DMA of step k+1 in flight
// under the MFMAs of step k. LDS rows are 128 B; logical 16-byte chunk c of row r lives at physical
// chunk (c ^ ((r >> 1) & 7)): the DMA image is lane-linear, so the permutation is applied to each
// lane's SOURCE address, and the same XOR on the ds_read_b128 fragment reads makes them conflict
// free (bank row = 256 B = two LDS rows). Zero padding: out-of-image taps read a 16-byte zero
// block kept at the end of every activation allocation (inside the descriptor's range).
// Epilogue: accumulators -> LDS as f32 [m][ch] (lane holds 4 consecutive channels per register
// quad), then whole 16-byte f16 channel groups are written with coalesced row stores.
#include "yh_internal.h"

namespace yh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int A, int B>
struct cmax { static constexpr int v = A > B ? A : B; };

// STAGES == 2: double buffer, __syncthreads() per step (drains the DMA: vmcnt(0)).
// STAGES == 3: ring of three; the DMA of step k+2 is issued while step k computes, each step waits
//              only for ITS tile with a counted s_waitcnt vmcnt(N) (N = this wave's DMA instructions
//              per tile) and a raw s_barrier, so one tile stays in flight across every barrier.
// ML: multi-level input - the rows of an image are the cells of up to five pyramid levels laid end
//              to end; every staged row carries its own level's height and width, so a tap never
//              leaves the level (the shared prediction head as one launch over the whole pyramid).
// EPI: the epilogue's f32 staging image covers TM / EPI rows at a time (a 256 x 256 tile's does not
//      fit in LDS at once).
// Residual of output row m, channels ch .. ch + 7: a plain load, or (res_up) the bilinear resize of the lower-resolution
// tensor evaluated at this row's pixel, one IEEE operation per operator as in bilinear_f16 (elementwise.hip; this file is
// built with contraction on, hence the pragma), rounded to f16 like the tensor it replaces.
// RESUP is a template flag of the conv kernel: the bilinear path costs ~24 VGPRs while the accumulators are live, which
// took the streaming 1x1 tiles from 4 to 3 waves per SIMD (-10 % on 25 launches) when it was a run-time branch.
__device__ __forceinline__ half8 bilinear_residual(const ConvParams& p, int m, int ch);
template <bool RESUP>
__device__ __forceinline__ half8 load_residual(const ConvParams& p, int m, int ch, long long ro) {
    if constexpr (RESUP) { if (p.res_up) return bilinear_residual(p, m, ch); }
    return *(const half8*)(p.res + ro);
}
__device__ __forceinline__ half8 bilinear_residual(const ConvParams& p, int m, int ch) {
#pragma clang fp contract(off)
    const int PQ = p.P * p.Q, n = m / PQ, rem = m - n * PQ, oy = rem / p.Q, ox = rem - oy * p.Q;
    const float sy = (float)p.res_h / (float)p.P, sx = (float)p.res_w / (float)p.Q;
    float fy = ((float)oy + 0.5f) * sy - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    float fx = ((float)ox + 0.5f) * sx - 0.5f;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, y1 = y0 + 1 < p.res_h ? y0 + 1 : p.res_h - 1;
    const int x0 = (int)fx, x1 = x0 + 1 < p.res_w ? x0 + 1 : p.res_w - 1;
    const float ly = fy - (float)y0, hy = 1.0f - ly, lx = fx - (float)x0, hx = 1.0f - lx;
    const half_t* rb = p.res + n * p.res_img_stride + ch;
    const half8 p00 = *(const half8*)(rb + ((long long)y0 * p.res_w + x0) * p.ldres);
    const half8 p01 = *(const half8*)(rb + ((long long)y0 * p.res_w + x1) * p.ldres);
    const half8 p10 = *(const half8*)(rb + ((long long)y1 * p.res_w + x0) * p.ldres);
    const half8 p11 = *(const half8*)(rb + ((long long)y1 * p.res_w + x1) * p.ldres);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float top = hx * (float)p00[e] + lx * (float)p01[e];
        const float bot = hx * (float)p10[e] + lx * (float)p11[e];
        o[e] = (half_t)(hy * top + ly * bot);
    }
    return o;
}

template <int MT, class ACC>
__device__ __forceinline__ ACC mfma_f16(const half8 a, const half8 b, const ACC c) {
    if constexpr (MT == 32) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// SPLITK: the grid also splits K into p.k_slices ranges; each workgroup writes its raw f32 partial
//      tile to p.partial[slice][m][ch] and splitk_reduce_f16 sums the slices in order (fixed order:
//      bitwise reproducible), adds bias/residual, activates and rounds. For launches whose M gives
//      only a handful of tiles (batch 1, deep layers: 12 workgroups streaming 4.7 MB of weights).
// MT:  MFMA tile edge. 32: v_mfma_f32_32x32x16_f16 (K = 16 per instruction, 16 accumulators per
//      lane); 16: v_mfma_f32_16x16x32_f16 (K = 32, 4 accumulators per lane: lane l holds output
//      pixel l & 15 and channels 4 (l >> 4) .. + 3). Same FLOPs per cycle; on real data the chip
//      holds a higher clock on the 16x16x32 shape (guide: DVFS give-back item 7).
// DUAL: two-source 1x1 form (ConvParams::x2): after k-step k1steps - 1 the loader switches to the second tensor - its
//      per-row source offsets are recomputed IN PLACE at the switch (no registers of their own while the accumulators
//      are live) and the buffer descriptor is swapped; the weight panel simply continues along K.
// TAIL: fused 1x1 tail (ConvParams::w2): the epilogue rounds the accumulators (bias, ReLU) to f16 STRAIGHT into an LDS
//      image [256 rows][256 ch] (128 KB, the dead ring; 16-byte chunk c of row r at chunk c ^ (r & 15): the lanes' 8-byte
//      writes and the 16-lane fragment reads are both conflict free) - no f32 staging, no rounds - and the eight waves
//      run the 32 x 256 second convolution on it with v_mfma_f32_16x16x32_f16 (weights = A straight from L2: 16 KB in
//      all). The first conv's output never leaves the CU.
template <int TCH, int TM, int WCH, int WM, bool SMALLC, int STAGES, int EPI, bool SPLITK = false, int MT = 32, bool ML = false, bool FP8 = false, bool RESUP = false, bool DUAL = false, bool TAIL = false, bool K3 = false>
__global__ __launch_bounds__(WCH * WM * 64, 2) void conv_igemm_f16(const ConvParams p) {
    static_assert(!ML || !SMALLC, "multi-level input: ordinary channel counts only");
    static_assert(!TAIL || (TCH == 256 && TM == 256 && WCH * WM == 8 && MT == 16 && !SPLITK && !RESUP && !DUAL && !ML), "fused 1x1 tail: the 256 x 256 tile of 16x16x32 MFMAs");
    static_assert(!DUAL || (!ML && !SMALLC && !FP8 && !RESUP), "two-source form: plain 1x1 convolutions");
    // K3: the 3x3 specialisation of the streaming tile (kernel extent known at compile time) - a kernel symbol of its own,
    // so that the MFMA-leaning 3x3 launches and the HBM-bound 1x1 launches of that tile are told apart in every profile
    const int kR = K3 ? 3 : p.R, kS = K3 ? 3 : p.S;
    constexpr int NW = WCH * WM, NT = NW * 64, RSTEP = NW * 8;  // waves, threads, rows per DMA pass
    constexpr int WTC = TCH / WCH, WTM = TM / WM;  // wave tile
    constexpr int TC = WTC / MT, TMT = WTM / MT;   // MFMA tiles per wave
    constexpr int KS = MT == 32 ? 4 : 2;           // MFMA k-slices per 64-deep step
    constexpr int NACC = MT == 32 ? 16 : 4;        // accumulator registers per MFMA tile
    typedef float accv __attribute__((ext_vector_type(NACC)));
    constexpr int XL = TM / RSTEP, WL = TCH / RSTEP;  // LDS-DMA instructions per thread per tile
    constexpr int AB_BYTES = (TCH + TM) * 128;
    constexpr int ES = TCH + 4;                    // epilogue row stride in floats
    constexpr int RING_BYTES = STAGES * AB_BYTES;
    constexpr int LDS_BYTES = cmax<RING_BYTES, TAIL ? TM * 512 : (TM / EPI) * ES * 4>::v;
    static_assert(WM % EPI == 0, "epilogue split");
    static_assert((NW == 4 || NW == 8) && WTC % MT == 0 && WTM % MT == 0 && (MT == 32 || MT == 16) && TM % RSTEP == 0 && TCH % RSTEP == 0, "tile shape");
    static_assert(STAGES == 1 || STAGES == 2 || (STAGES == 3 && !SMALLC), "ring variants: no ordinary loads may share the loop");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    // XCD-aware bijective remap: consecutive work ids (which share activation rows / weight
    // panels) land on one XCD's L2 instead of being dealt round-robin over the eight.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int tile_id = wg, kslice = 0, kt0 = 0, nk_total = p.ksteps;
    if (SPLITK) {  // slice is the slow index: neighbours share the K range (weights / activations in L2)
        const int ntiles = nwg / p.k_slices;
        kslice = wg / ntiles;
        tile_id = wg - kslice * ntiles;
        kt0 = kslice * p.ksteps_per_slice;
        nk_total = p.ksteps - kt0 < p.ksteps_per_slice ? p.ksteps - kt0 : p.ksteps_per_slice;
    }
    const int ch_tile = tile_id % p.n_ch_tiles + p.ch_tile0, m_tile = tile_id / p.n_ch_tiles + p.m_tile0;

    const int tid = threadIdx.x, chunk = tid & 7, rb = tid >> 3;
    const int PQ = p.P * p.Q;

    // ---- per-thread staging rows of the activation tile.
    // LDS-DMA (buffer_load ... lds) writes wave-uniform base + lane*16: thread tid always fills
    // physical 16-byte chunk (tid & 7) of row (tid >> 3) + 32 i, so the XOR swizzle moves to the
    // SOURCE side: it fetches logical chunk lc = (tid & 7) ^ ((row >> 1) & 7).
    const int lc = chunk ^ ((rb >> 1) & 7);
    int xbase[XL], xih[XL], xiw[XL];
    int xH[ML ? XL : 1], xW[ML ? XL : 1];   // ML: the row's own level geometry (stride 1 only)
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int m = m_tile * TM + rb + RSTEP * i;
        if (ML) { xH[i] = 1; xW[i] = 1; }
        if (m < p.M) {
            const int n = m / PQ, rem = m - n * PQ;
            if (ML) {
                int st = 0, hh = p.lev_h[0], ww = p.lev_w[0];
#pragma unroll
                for (int L = 1; L < 5; ++L)
                    if (L < p.nlev && rem >= p.lev_start[L]) { st = p.lev_start[L]; hh = p.lev_h[L]; ww = p.lev_w[L]; }
                const int local = rem - st, op = local / ww, oq = local - op * ww;
                xih[i] = op - p.pad;
                xiw[i] = oq - p.pad;
                xH[i] = hh; xW[i] = ww;
                xbase[i] = (int)(n * p.x_img_stride) + (st + xih[i] * ww + xiw[i]) * p.C + lc * 8;
                continue;
            }
            const int op = rem / p.Q, oq = rem - op * p.Q;
            xih[i] = op * p.stride - p.pad;
            xiw[i] = oq * p.stride - p.pad;
            xbase[i] = (int)(n * p.x_img_stride) + (xih[i] * p.W + xiw[i]) * p.C + (SMALLC ? 0 : lc * 8);
        } else {
            xih[i] = -(1 << 24);
            xiw[i] = 0;
            xbase[i] = 0;
        }
    }
    __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    unsigned x_zero_off = p.x_zero_off;
    int wk_shift = 0;   // DUAL: K index at which the current source's channels start in the weight panel
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const unsigned wbase = (unsigned)(((p.ch_base + ch_tile * TCH + rb) * p.ldw + lc * 8) * 2);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds3 = (lds_char*)lds;

    // position of the NEXT tile to load along K
    int kr = 0, ks = 0, kc = 0, kt_load = kt0;
    if (SPLITK && !SMALLC) {
        const int taps = kR * kS, cb = kt0 / taps, rs = kt0 - cb * taps;
        kc = cb << 6;
        kr = rs / kS;
        ks = rs - kr * kS;
    }
    // DUAL: from k-step k1steps on, the rows come from the second tensor (1x1, no padding: a row is in the image or past M)
    auto second_source = [&]() {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int m = m_tile * TM + rb + RSTEP * i;
            if (m < p.M) {
                const int n = m / PQ, rem = m - n * PQ, op = rem / p.Q, oq = rem - op * p.Q;
                xbase[i] = (int)(n * p.x2_img_stride) + (op * p.W2 + oq) * p.stride2 * p.C2 + lc * 8;
                xih[i] = 0;
            }
            xiw[i] = 0;
        }
        xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, (int)p.x2_bytes, 0x00020000);
        x_zero_off = p.x2_zero_off;
        wk_shift = p.k1steps << 6;
        kc = (kt_load - p.k1steps) << 6;
    };
    if (DUAL && kt_load >= p.k1steps) second_source();   // (a split-K slice that starts inside the second source)

    // One tile's DMA is NDMA instructions per thread (WL weight pieces, then XL activation pieces).
    // tile_begin fixes the tile's K position, tile_part issues piece d, tile_end advances K.
    constexpr int NDMA = XL + WL;
    int t_r = 0, t_s = 0, t_off = 0, t_wk = 0, kc_cur = 0;
    auto tile_begin = [&]() {
        if (SMALLC) {
            const int2 tap = p.rs_table[kt_load * 8 + lc];
            t_r = tap.x; t_s = tap.y;
            t_off = (t_r * p.W + t_s) * p.C;   // chunk = 16 bytes starting at tap (r,s): 8/C pixels
            t_wk = kt_load * 64;
        } else {
            t_r = kr; t_s = ks; kc_cur = kc;
            t_off = (kr * p.W + ks) * p.C + kc;
            t_wk = (kr * kS + ks) * p.C + kc;   // K index of this step in the [(r,s,c)] weight panel
            if (DUAL) t_wk = wk_shift + kc;
        }
    };
    auto tile_part = [&](int buf, int d) {
        if (p.skip_dma) return;
        lds_char* const dstw = lds3 + buf * AB_BYTES + wave * 1024;
        if (d < WL) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, dstw + d * (RSTEP * 128), 16,
                                                     (int)(wbase + (unsigned)((RSTEP * d) * p.ldw + t_wk) * 2u), 0, 0, 0);
        } else {
            const int i = d - WL;
            const bool ok = ML ? ((unsigned)(xih[i] + t_r) < (unsigned)xH[i] && (unsigned)(xiw[i] + t_s) < (unsigned)xW[i])
                               : ((unsigned)(xih[i] + t_r) < (unsigned)p.H && (unsigned)(xiw[i] + t_s) < (unsigned)p.W);
            // padded taps read the 16-byte zero block that ends every activation allocation
            const int toff = ML ? (t_r * xW[i] + t_s) * p.C + kc_cur : t_off;
            const unsigned voff = ok ? (unsigned)(xbase[i] + toff) * 2u : x_zero_off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, dstw + TCH * 128 + i * (RSTEP * 128), 16, (int)voff, 0, 0, 0);
        }
    };
    auto tile_end = [&]() {
        ++kt_load;
        if (!SMALLC) {
            // K order: 64-channel chunk OUTER, taps INNER. The R*S taps of one chunk read the same
            // input lines (shifted by one pixel / one row) in consecutive steps, so per XCD the
            // live set is ~(tile pixels + halo) * 128 B * 32 CUs + the chunk's weights: it fits
            // the 4 MiB L2, where tap-outer order streamed ~8 MB between two uses of a line.
            if (++ks == kS) { ks = 0; if (++kr == kR) { kr = 0; kc += 64; } }
        }
        if (DUAL && kt_load == p.k1steps) second_source();
    };
    auto load_tile = [&](int buf) {
        tile_begin();
#pragma unroll
        for (int d = 0; d < NDMA; ++d) tile_part(buf, d);
        tile_end();
    };

    const int lane = tid & 63, wid = tid >> 6;
    const int wc = wid / WM, wm = wid % WM;
    // fragment lane map: row (channel / pixel) lr of the MFMA tile, 16-byte k-group lh within a k-slice
    const int l31 = lane & (MT - 1), lh = MT == 32 ? lane >> 5 : lane >> 4, swz = (l31 >> 1) & 7;
    const int a_row = (wc * WTC + l31) * 128, b_row = TCH * 128 + (wm * WTM + l31) * 128;
    constexpr int TSTR = MT * 128;                 // LDS bytes between consecutive MFMA tiles' rows
    constexpr int KG = MT == 32 ? 2 : 4;           // 16-byte k-groups per k-slice

    accv acc[TC][TMT];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TMT; ++j)
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[i][j][e] = 0.0f;

    // ---- epilogue geometry is known up front; for single-round epilogues the residual rows are
    // requested NOW, so their HBM latency hides under the whole main loop (the residual 1x1 convs
    // have 1-4 k-steps: requested after the loop, that latency was fully exposed).
    // (lanes per output row: TCH / 8, rounded up to a power of two - the 96-channel tile keeps 12 of 16 busy)
    constexpr int TPR = TCH / 8 <= 4 ? 4 : (TCH / 8 <= 8 ? 8 : (TCH / 8 <= 16 ? 16 : 32)), RPP = NT / TPR, EROWS = TM / EPI, NPASS = EROWS / RPP, WMG = WM / EPI;
    static_assert(TCH / 8 <= 32 && NT % TPR == 0 && EROWS % RPP == 0, "epilogue geometry");
    const int ch_l = (tid % TPR) * 8, rr = tid / TPR;
    const int ch = p.ch_base + ch_tile * TCH + ch_l;
    const bool ch_ok = ch < p.cout8 && ch_l < TCH;
    auto offsets = [&](int m, long long& yo, long long& ro) {
        if (p.y_dense) {
            yo = (long long)m * p.ldy + ch;
            ro = (long long)m * p.ldres + ch;
        } else {
            const int n = m / PQ, rem = m - n * PQ;
            yo = n * p.y_img_stride + (long long)rem * p.ldy + ch;
            ro = n * p.res_img_stride + (long long)rem * p.ldres + ch;
        }
    };
    half8 rv0[EPI == 1 ? NPASS : 1];
    if (EPI == 1 && !SPLITK && p.res && ch_ok) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int m = m_tile * TM + pass * RPP + rr;
            long long yo, ro;
            offsets(m < p.M ? m : 0, yo, ro);
            rv0[pass] = load_residual<RESUP>(p, m < p.M ? m : 0, ch, ro);
        }
    }

    if constexpr (STAGES == 2 && FP8) {
        // fp8 (OCP E4M3) operands, block-scaled MFMA with unit scales: a 128-byte LDS row is 128 channels, one
        // v_mfma_scale_f32_16x16x128_f8f6f4 consumes it whole (lane group g = lane >> 4 feeds bytes
        // 32 g .. 32 g + 31 of its row for A and for B alike: the instruction pairs equal (g, byte)).
        // A fragments are read in two halves so that 8 + 4 operand tiles of 8 VGPRs do not all live at once.
        static_assert(MT == 16 && !SMALLC && !SPLITK && TC % 2 == 0, "fp8 form: 16x16x128, ordinary channels");
        typedef int v8i __attribute__((ext_vector_type(8)));
        auto read_op = [&](const char* row) {
            const u32x4 lo = *(const u32x4*)(row + (((2 * lh) ^ swz) << 4)), hi = *(const u32x4*)(row + (((2 * lh + 1) ^ swz) << 4));
            v8i v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (int)lo[e]; v[4 + e] = (int)hi[e]; }
            return v;
        };
        load_tile(0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk_total; ++kt) {
            const bool more = kt + 1 < nk_total;
            const char* base = lds + cur * AB_BYTES;
            if (more) tile_begin();
            v8i b[TMT];
#pragma unroll
            for (int j = 0; j < TMT; ++j) b[j] = read_op(base + b_row + j * TSTR);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                v8i a[TC / 2];
#pragma unroll
                for (int i = 0; i < TC / 2; ++i) a[i] = read_op(base + a_row + (hh * (TC / 2) + i) * TSTR);
                if (more) {
#pragma unroll
                    for (int d = (hh * NDMA) / 2; d < ((hh + 1) * NDMA) / 2; ++d) tile_part(cur ^ 1, d);
                }
#pragma unroll
                for (int i = 0; i < TC / 2; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j)
                        acc[hh * (TC / 2) + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[hh * (TC / 2) + i][j], 0, 0, 0, 127, 0, 127);
            }
            if (more) tile_end();
            __syncthreads();
            cur ^= 1;
        }
    } else if (STAGES == 1) {
        // One LDS stage, no overlap inside the workgroup: for the HBM-bound 1x1 layers (K <= 256, one to
        // four steps) what hides latency is the number of workgroups per CU, and 34 KB of LDS (with the
        // split epilogue) lets four of them live on a CU instead of two.
        for (int kt = 0; kt < nk_total; ++kt) {
            if (kt) __syncthreads();   // every wave is done reading the previous tile
            load_tile(0);
            __syncthreads();           // vmcnt(0) + barrier: the tile has landed and is visible
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int co = ((KG * kk + lh) ^ swz) << 4;
                half8 a[TC], b[TMT];
#pragma unroll
                for (int i = 0; i < TC; ++i) a[i] = *(const half8*)(lds + a_row + i * TSTR + co);
#pragma unroll
                for (int j = 0; j < TMT; ++j) b[j] = *(const half8*)(lds + b_row + j * TSTR + co);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j) acc[i][j] = mfma_f16<MT>(a[i], b[j], acc[i][j]);
            }
        }
        __syncthreads();  // LDS is reused by the epilogue
    } else if (STAGES == 2) {
        // Double buffer. The DMA of step k+1 is issued in four slices, one per 16-wide k-slice of
        // step k, between that slice's fragment reads and its MFMAs: DMA issue (the expensive
        // part of LDS-DMA for the issuing wave) overlaps MFMA execution instead of preceding it.
        load_tile(0);
        __syncthreads();  // drains the DMA (vmcnt(0)) and publishes tile 0
        int cur = 0;
        for (int kt = 0; kt < nk_total; ++kt) {
            const bool more = kt + 1 < nk_total;
            const char* base = lds + cur * AB_BYTES;
            if (more) tile_begin();
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int co = ((KG * kk + lh) ^ swz) << 4;
                half8 a[TC], b[TMT];
#pragma unroll
                for (int i = 0; i < TC; ++i) a[i] = *(const half8*)(base + a_row + i * TSTR + co);
#pragma unroll
                for (int j = 0; j < TMT; ++j) b[j] = *(const half8*)(base + b_row + j * TSTR + co);
                if (more) {
#pragma unroll
                    for (int d = (kk * NDMA) / KS; d < ((kk + 1) * NDMA) / KS; ++d) tile_part(cur ^ 1, d);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j) acc[i][j] = mfma_f16<MT>(a[i], b[j], acc[i][j]);
            }
            if (more) tile_end();
            __syncthreads();  // all waves done with buf[cur]; DMA into buf[cur^1] landed
            cur ^= 1;
        }
    } else {
        // 3-stage ring + register double buffering of the MFMA fragments:
        //   step k:  wait(tile k+1 landed) ; lgkmcnt(0) ; barrier ; DMA tile k+3 -> stage of tile k ;
        //            ds_read all fragments of tile k+1 -> Rnext ; 16 MFMAs on Rcur (tile k)
        // so the MFMAs of a step never wait on LDS latency (their operands were read one step
        // earlier) and every DMA has two steps to land. Stage k is free for refill after the
        // barrier because every wave drained its reads of tile k (lgkmcnt(0)) before arriving.
        struct Frag { half8 a[KS][TC], b[KS][TMT]; };
        auto read_frags = [&](int buf, Frag& f) {
            const char* base = lds + buf * AB_BYTES;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int co = ((KG * kk + lh) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < TC; ++i) f.a[kk][i] = *(const half8*)(base + a_row + i * TSTR + co);
#pragma unroll
                for (int j = 0; j < TMT; ++j) f.b[kk][j] = *(const half8*)(base + b_row + j * TSTR + co);
            }
        };
        auto mfma_frags = [&](const Frag& f) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TMT; ++j) acc[i][j] = mfma_f16<MT>(f.a[kk][i], f.b[kk][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        const int nk = nk_total;
        constexpr int RS = STAGES;   // ring size (3 or 4): RS - 1 tiles in flight after the prologue, RS - 2 across a step
        // waits until all but the `younger` most recently issued tiles have landed (compile-time immediates)
        auto wait_younger = [&](int younger) {
            if (younger >= 2 && RS >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#pragma unroll
        for (int t = 0; t < RS; ++t)
            if (t < nk) load_tile(t);
        // tile 0 landed: at most min(nk, RS) - 1 younger tiles may stay in flight
        {
            const int younger = (nk < RS ? nk : RS) - 1;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else wait_younger(younger);
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        Frag f0, f1;
        read_frags(0, f0);
        int st1 = 1, st0 = 0;  // stage of tile kt+1, stage of tile kt (refilled with tile kt+RS)
        auto step = [&](int kt, Frag& cur, Frag& nxt) {
            if (kt + 1 < nk) {  // tile kt+1 landed; tiles kt+2 .. kt+RS-1 (those that exist) may still be in flight
                const int rest = nk - kt - 2;
                wait_younger(rest < RS - 2 ? rest : RS - 2);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of tile kt are in registers
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + RS < nk) load_tile(st0);
            if (kt + 1 < nk) read_frags(st1, nxt);
            mfma_frags(cur);
            st0 = st0 == RS - 1 ? 0 : st0 + 1;
            st1 = st1 == RS - 1 ? 0 : st1 + 1;
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            step(kt, f0, f1);
            step(kt + 1, f1, f0);
        }
        if (kt < nk) step(kt, f0, f1);
        __syncthreads();  // LDS is reused by the epilogue
    }

    if constexpr (TAIL) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        typedef float acc4 __attribute__((ext_vector_type(4)));
        // (the main loop's last barrier has passed: no wave reads the ring any more)
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int c_l = wc * WTC + i * MT + 4 * lh;
            const f32x4 b4 = *(const f32x4*)(p.bias + c_l);
            f32x4 s4 = { 1.0f, 1.0f, 1.0f, 1.0f };
            if (FP8) s4 = *(const f32x4*)(p.scale + c_l);
#pragma unroll
            for (int j = 0; j < TMT; ++j) {
                const int m_l = wm * WTM + j * MT + l31;
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = FP8 ? __builtin_fmaf(acc[i][j][e], s4[e], b4[e]) : acc[i][j][e] + b4[e];
                    if (p.act == 1) v = fmaxf(v, 0.0f);
                    o[e] = (half_t)v;
                }
                *(half4*)(lds + m_l * 512 + (((c_l >> 3) ^ (m_l & 15)) << 4) + (c_l & 4) * 2) = o;
            }
        }
        __syncthreads();
        // second convolution: wave w -> rows 32 w .. 32 w + 31 (two 16-row pixel tiles), both 16-channel tiles
        const int w8 = tid >> 6;
        half8 af[2][8];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) af[ct][k2] = *(const half8*)(p.w2 + (ct * 16 + l31) * 256 + k2 * 32 + lh * 8);
        const f32x4 bb0 = *(const f32x4*)(p.bias2 + 4 * lh), bb1 = *(const f32x4*)(p.bias2 + 16 + 4 * lh);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int row = w8 * 32 + pt * 16 + l31;
            const char* yrow = lds + row * 512;
            acc4 a0 = { 0.0f, 0.0f, 0.0f, 0.0f }, a1 = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const half8 bf = *(const half8*)(yrow + (((k2 * 4 + lh) ^ l31) << 4));
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][k2], bf, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][k2], bf, a1, 0, 0, 0);
            }
            const int m2 = m_tile * TM + row;   // lane: pixel `row`, channels 4 lh .. + 3 of each 16-channel tile
            if (m2 < p.M) {
                half4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] = (half_t)fmaxf(a0[e] + bb0[e], 0.0f); o1[e] = (half_t)fmaxf(a1[e] + bb1[e], 0.0f); }
                *(half4*)(p.y2 + (long long)m2 * 32 + 4 * lh) = o0;
                *(half4*)(p.y2 + (long long)m2 * 32 + 16 + 4 * lh) = o1;
            }
        }
        return;
    }
    // ---- epilogue: accumulators -> LDS f32 [m][ch] (TM / EPI rows per round), then coalesced rows
    float* E = (float*)lds;
    float bias8[8];
    {
        const f32x4 b0 = *(const f32x4*)(p.bias + ch), b1 = *(const f32x4*)(p.bias + ch + 4);  // bias is padded to coutPad
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    // (fp8 precision, a tensor that feeds an E4M3 convolution: its channels' reciprocal scales, once per tile - read inside the row
    // loop they were re-loaded behind every store)
    f32x4 inv0 = { 0.0f, 0.0f, 0.0f, 0.0f }, inv1 = inv0;
    if (p.y8 && ch_ok) { inv0 = *(const f32x4*)(p.y8_inv + ch); inv1 = *(const f32x4*)(p.y8_inv + ch + 4); }
    float scale8[FP8 ? 8 : 1];
    if (FP8) {
        const f32x4 s0 = *(const f32x4*)(p.scale + ch), s1 = *(const f32x4*)(p.scale + ch + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { scale8[e] = s0[e]; scale8[4 + e] = s1[e]; }
    }
#pragma unroll
    for (int h = 0; h < EPI; ++h) {
        // multi-round epilogues: the residual rows of this round are requested up front (one latency)
        half8 rv[NPASS];
        if (EPI == 1) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) rv[pass] = rv0[pass];
        } else if (!SPLITK && p.res && ch_ok) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m = m_tile * TM + h * EROWS + pass * RPP + rr;
                long long yo, ro;
                offsets(m < p.M ? m : 0, yo, ro);
                rv[pass] = load_residual<RESUP>(p, m < p.M ? m : 0, ch, ro);
            }
        }
        if (EPI == 1 || wm / WMG == h) {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TMT; ++j) {
                    const int m_l = (wm % WMG) * WTM + j * MT + l31;
                    // 32x32: register quad g holds channels 8 g + 4 (lane >> 5) .. + 3; 16x16: the one
                    // quad holds channels 4 (lane >> 4) .. + 3 (C/D maps of the guide, weights = A operand)
#pragma unroll
                    for (int g = 0; g < NACC / 4; ++g) {
                        const int c_l = wc * WTC + i * MT + (MT == 32 ? 8 * g + 4 * lh : 4 * lh);
                        f32x4 v = { acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3] };
                        *(f32x4*)(E + m_l * ES + c_l) = v;
                    }
                }
        }
        __syncthreads();
        if (SPLITK) {   // the raw f32 partial tile -> this K slice's slab; splitk_reduce_f16 sums the slabs in slice order and runs the epilogue
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m_l = pass * RPP + rr, m = m_tile * TM + h * EROWS + m_l;
                if (m < p.M) {
                    float* dst = p.partial + ((long long)kslice * p.M + m) * p.partial_ld + ch;
                    *(f32x4*)dst = *(const f32x4*)(E + m_l * ES + ch_l);
                    *(f32x4*)(dst + 4) = *(const f32x4*)(E + m_l * ES + ch_l + 4);
                }
            }
            return;
        }
        if (ch_ok) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m_l = pass * RPP + rr, m = m_tile * TM + h * EROWS + m_l;
                if (m < p.M) {
                    const f32x4 v0 = *(const f32x4*)(E + m_l * ES + ch_l), v1 = *(const f32x4*)(E + m_l * ES + ch_l + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (FP8) { v[e] = __builtin_fmaf(v0[e], scale8[e], bias8[e]); v[4 + e] = __builtin_fmaf(v1[e], scale8[4 + e], bias8[4 + e]); }
                        else { v[e] = v0[e] + bias8[e]; v[4 + e] = v1[e] + bias8[4 + e]; }
                    }
                    long long yo, ro;
                    offsets(m, yo, ro);
                    if (p.res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)rv[pass][e];
                    }
                    if (p.act == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if (ch + 8 > p.tanh_from) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ch + e >= p.tanh_from) v[e] = spec_tanhf(v[e]);
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    if (p.y) *(half8*)(p.y + yo) = o;
                    if (p.y8) {   // fp8 precision: this tensor feeds an fp8 convolution (quantised from the f16-rounded value)
                        const unsigned lo = e4m3_pack4((float)o[0] * inv0[0], (float)o[1] * inv0[1], (float)o[2] * inv0[2], (float)o[3] * inv0[3]);
                        const unsigned hi = e4m3_pack4((float)o[4] * inv1[0], (float)o[5] * inv1[1], (float)o[6] * inv1[2], (float)o[7] * inv1[3]);
                        *(uint2*)(p.y8 + yo) = make_uint2(lo, hi);
                    }
                }
            }
        }
        if (h + 1 < EPI) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// stem_pool_f16 — the stem (7x7 stride 2, 3 -> 64, bias, ReLU) fused with the 3x3 stride-2 max pool.
// Unfused, the stem writes 64 channels at 275 x 275 per frame (9.7 MB) only for the pool to read
// them back and keep a quarter; fused, the block keeps its stem pixels in LDS and only the pooled
// tensor goes to HBM. The implicit GEMM also stops paying for im2col: the block stages the INPUT
// patch once (39 x 40 pixels of 8 bytes for 17 x 17 stem pixels) and every B fragment is one
// aligned ds_read_b128 out of it - k-step r of the 16x16x32 MFMA is kernel row r, its 32 K values
// the 8 pixel columns x 4 stored channels right of (2 sy + r, 2 sx), which are contiguous in the
// patch (the 8th column and the 4th channel carry zero weights).
// One workgroup = 8 x 8 pooled pixels = 17 x 17 stem pixels (one halo row and column recomputed:
// 13 % more MFMAs), 8 waves = 4 walkers over the 16-pixel groups x 2 channel halves; each wave keeps
// its weight fragments (2 channel tiles x 7 rows) in registers for the whole launch; the grid is
// persistent.
// Border: stem pixels outside the image are staged as 0, which max() ignores because every real
// stem pixel is >= 0 after the ReLU (the pool's padding is "absent", as in the oracle).
// ------------------------------------------------------------------------------------------------
#define SP_PT 8                 // pooled tile edge
#define SP_ST (2 * SP_PT + 1)   // stem tile edge (17)
#define SP_NPX (SP_ST * SP_ST)  // 289
#define SP_PR (2 * SP_ST + 5)   // input patch rows (39)
#define SP_PC 42                // input patch columns (40 used), 8 bytes each
#define SP_SS 184               // staging row stride in bytes (64 ch f16 = 128 + 56)
// LDS images are laid out against bank conflicts by the lane-group rules of ds_read_b128 / ds_write_b64 (tools/study/
// stem_lds_banks.py counts them): the 19 groups of 16 stem pixels are the 17 tile rows (columns 0 .. 15) plus column 16 in two
// groups, so a group never wraps over a row; with 42 patch columns (row pitch 336 B) and a 184-byte staging pitch every B
// fragment read and every staging write is conflict free and the pool's reads are 2-way (before: 289 pixels taken 16 at a
// time, pitches 40 / 144 - 1.8-way, 1.9-way and 3-way; SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE).
__global__ __launch_bounds__(512, 4) void stem_pool_f16(const StemPoolParams p) {   // (4 waves per SIMD = two workgroups per CU: at most 128 VGPRs)
    __shared__ __attribute__((aligned(16))) char patch[SP_PR * SP_PC * 8];
    __shared__ __attribute__((aligned(16))) char stage[SP_NPX * SP_SS];
    // Round 5: what a lane needs to know about pixel group mt - where its B fragment starts in the patch, where its stem pixel sits in the
    // staging image, whether it holds a pixel at all - depends on (mt, lane) only, not on the tile: tabulated once per workgroup (9.7 KB)
    // instead of ~17 vector instructions of selects and multiplies per group and tile. word 0: patch byte offset | mi << 16 | mj << 21 |
    // in_tile << 26; word 1: staging byte offset of the pixel (+ this lane's channel quad).
    __shared__ uint2 gtab[(SP_ST + 2) * 64];
    typedef float accv __attribute__((ext_vector_type(4)));
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lh = lane >> 4;
    const int wp = wave >> 1, ct0 = (wave & 1) * 2;   // 8 waves: 4 pixel-group walkers x 2 channel halves

    half8 a[2][7];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 7; ++r) a[ct][r] = *(const half8*)(p.w + ((ct0 + ct) * 16 + l15) * 256 + r * 32 + lh * 8);
    // (the 64 biases live in LDS, not in 8 registers per lane: with the group table's two words the kernel stood at 130 VGPRs - one
    // workgroup per CU instead of two; a group's accumulators start from two ds_read_b128 that land under its first B-fragment read)
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    if (tid < 64) bias_s[tid] = p.bias[tid];

    for (int e = tid; e < (SP_ST + 2) * 64; e += 512) {
        const int mt = e >> 6, ln = e & 63, c15 = ln & 15, q = ln >> 4;
        const int in_t = mt < SP_ST + 1 || c15 == 0;
        const int mi = mt < SP_ST ? mt : (mt == SP_ST ? c15 : SP_ST - 1), mj = mt < SP_ST ? c15 : SP_ST - 1;
        gtab[e] = make_uint2((unsigned)(((2 * mi) * SP_PC + 2 * mj + 2 * q) * 8) | ((unsigned)mi << 16) | ((unsigned)mj << 21) | ((unsigned)in_t << 26),
                             (unsigned)((mi * SP_ST + mj) * SP_SS + 4 * q * 2));
    }
    const int tiles_img = p.tiles_y * p.tiles_x, total = p.n * tiles_img;
    constexpr int NCH = SP_PR * (SP_PC / 2), NLD = (NCH + 511) / 512;   // 16-byte patch chunks, per-thread slots

    half8 nv[NLD];
    unsigned raw[NLD][2];   // rgb form: two pixels' raw bytes (c0 | c1 << 8 | c2 << 16 | valid << 24) - converted in store_patch
    // the next tile's patch travels through registers: loads are issued before this tile's MFMA
    // phase and land in LDS after it, so their latency hides under the compute (the rgb form keeps the RAW bytes over the
    // MFMA phase and normalises them in store_patch)
    auto fetch_patch = [&](int tile) {
        const int b = tile / tiles_img, rem = tile - b * tiles_img;
        const int by0 = 4 * ((rem / p.tiles_x) * SP_PT) - 2, bx0 = 4 * ((rem % p.tiles_x) * SP_PT) - 2;   // even column
        const half_t* img = p.x + (long long)b * p.x_img_stride;
        if (p.rgb && by0 - 3 >= 0 && by0 - 3 + SP_PR <= p.S && bx0 - 3 >= 0 && bx0 - 3 + 40 <= p.S) {
            // INTERIOR tile (a wave-uniform test: 4 of 5 tiles at 550 x 550): the whole 39 x 40-pixel patch lies inside the image, so no
            // pixel needs a bounds test and a pixel pair is six consecutive bytes from a scalar base + a per-thread constant - one dword and
            // one short (unaligned: the target's global loads allow it) instead of four byte / short loads behind two branches
            struct __attribute__((packed, aligned(1))) U32 { unsigned v; };
            struct __attribute__((packed, aligned(1))) U16 { unsigned short v; };
            const uint8_t* base = p.rgb + (((long long)b * p.S + (by0 - 3)) * p.S + (bx0 - 3)) * 3;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + 512 * k, row = i / (SP_PC / 2), cp = i - row * (SP_PC / 2);
                raw[k][0] = raw[k][1] = 0u;
                if (i < NCH && cp < 20) {
                    const uint8_t* q = base + (row * p.S + 2 * cp) * 3;
                    raw[k][0] = ((const U32*)q)->v;          // (untouched until store_patch: the loads land under this tile's MFMA phase)
                    raw[k][1] = ((const U16*)(q + 4))->v | 0x80000000u;   // bit 31: the slot holds six packed bytes, not two flagged pixels
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 512 * k, row = i / (SP_PC / 2), cp = i - row * (SP_PC / 2);
            const int gy = by0 + row, gx = bx0 + 2 * cp;
            const bool used = i < NCH && cp < 20;   // (columns 40, 41 are pitch only)
            if (p.rgb) {   // fused preprocessing: two pixels of raw RGB, zero outside the image
                const int iy = gy - 3;
                raw[k][0] = raw[k][1] = 0u;
                if (used && (unsigned)iy < (unsigned)p.S) {
                    const uint8_t* row = p.rgb + ((long long)b * p.S + iy) * p.S * 3;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int ix = gx + px - 3;
                        if ((unsigned)ix < (unsigned)p.S)
                            raw[k][px] = (unsigned)row[ix * 3] | ((unsigned)row[ix * 3 + 1] << 8) | ((unsigned)row[ix * 3 + 2] << 16) | (1u << 24);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) nv[k][e] = (half_t)0.0f;
                if (used && (unsigned)gy < (unsigned)p.Hp && gx >= 0 && gx + 1 < p.Wp) nv[k] = *(const half8*)(img + ((long long)gy * p.Wp + gx) * 4);
            }
        }
    };
    auto store_patch = [&]() {
        // (v - mean) / std rounded to f16 (the preprocess kernel's expression) as (v - mean) * (1 / std): the same f16 for every
        // byte value of every channel (tests/test_stem_norm.py checks all 768), and no table read with its bank conflicts
        const float mean[3] = { 123.68f, 116.78f, 103.94f }, rs[3] = { 1.0f / 58.40f, 1.0f / 57.12f, 1.0f / 57.38f };
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 512 * k, row = i / (SP_PC / 2), cp = i - row * (SP_PC / 2);
            if (p.rgb) {
                if (raw[k][1] & 0x80000000u) {   // an interior tile's slot: bytes 0-3 | bytes 4-5 -> two flagged pixels
                    const unsigned lo = raw[k][0], hi = raw[k][1] & 0xFFFFu;
                    raw[k][0] = (lo & 0xFFFFFFu) | (1u << 24);
                    raw[k][1] = (lo >> 24) | (hi << 8) | (1u << 24);
                }
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const bool ok = (raw[k][px] >> 24) != 0u;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = (float)((raw[k][px] >> (8 * c)) & 255u);
                        nv[k][px * 4 + c] = ok ? (half_t)((v - mean[c]) * rs[c]) : (half_t)0.0f;
                    }
                    nv[k][px * 4 + 3] = (half_t)0.0f;
                }
            }
            if (i < NCH) *(half8*)(patch + (row * SP_PC + 2 * cp) * 8) = nv[k];
        }
    };
    if ((int)blockIdx.x < total) { fetch_patch(blockIdx.x); store_patch(); }
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = tile / tiles_img, rem = tile - b * tiles_img;
        const int py0 = (rem / p.tiles_x) * SP_PT, px0 = (rem % p.tiles_x) * SP_PT;
        const int next = tile + gridDim.x;
        __syncthreads();   // patch visible; the previous tile's pool phase is done with the staging image
        if (next < total) fetch_patch(next);
        // (a tile whose 17 x 17 stem pixels all lie inside the stem image - every tile off the image's border - needs no per-pixel range test)
        const bool inner = !p.stem && py0 > 0 && px0 > 0 && 2 * py0 + 15 < p.SO && 2 * px0 + 15 < p.SO;
        for (int mt = wp; mt < SP_ST + 2; mt += 4) {
            // group mt < 17: stem row mt, columns 0 .. 15; group 17: column 16, rows 0 .. 15; group 18: pixel (16, 16) - gtab
            const uint2 g = gtab[mt * 64 + lane];
            const bool in_tile = (g.x >> 26) & 1u;
            accv acc[2];   // start from the bias: one rounding fewer than (sum) + bias, and no separate add
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = *(const accv*)(bias_s + (ct0 + ct) * 16 + 4 * lh);
            const char* bp = patch + (g.x & 0xFFFFu);
            // (two fragments in flight: row r + 1 is requested before row r's MFMAs - written as a rotation of two named registers because
            // the compiler, left to itself at 127 VGPRs, waited for every read right behind its issue)
            half8 bf0 = *(const half8*)(bp), bf1 = *(const half8*)(bp + SP_PC * 8);
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const half8 bf = (r & 1) ? bf1 : bf0;
                if (r + 2 < 7) { if (r & 1) bf1 = *(const half8*)(bp + (r + 2) * SP_PC * 8); else bf0 = *(const half8*)(bp + (r + 2) * SP_PC * 8); }
                __builtin_amdgcn_sched_barrier(0);   // (the request stays in front of this row's MFMAs)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = mfma_f16<16>(a[ct][r], bf, acc[ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
            half4 zero4;
#pragma unroll
            for (int e = 0; e < 4; ++e) zero4[e] = (half_t)0.0f;
            char* const sp = stage + g.y + ct0 * 32;
            if (inner) {
                if (in_tile) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        // ReLU after the f16 rounding (rounding is monotonic and keeps 0): two packed ops
                        half4 o = __builtin_convertvector(acc[ct], half4);
                        o = __builtin_elementwise_max(o, zero4);
                        *(half4*)(sp + ct * 32) = o;
                    }
                }
            } else {
                const int mi = (int)((g.x >> 16) & 31u), mj = (int)((g.x >> 21) & 31u);
                const int sy = 2 * py0 - 1 + mi, sx = 2 * px0 - 1 + mj;
                const bool ok = in_tile && (unsigned)sy < (unsigned)p.SO && (unsigned)sx < (unsigned)p.SO;
                if (in_tile) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        half4 o = __builtin_convertvector(acc[ct], half4);
                        o = __builtin_elementwise_max(o, zero4);
                        o = ok ? o : zero4;
                        *(half4*)(sp + ct * 32) = o;
                        // test hook: each stem pixel is owned by the tile that holds it off the halo row / column
                        if (p.stem && ok && mi >= 1 && mj >= 1)
                            *(half4*)(p.stem + (long long)b * p.stem_img_stride + ((long long)sy * p.SO + sx) * 64 + (ct0 + ct) * 16 + 4 * lh) = o;
                    }
                }
            }
        }
        __syncthreads();
        if (next < total) store_patch();   // nobody reads the patch any more
        for (int wq = tid; wq < SP_PT * SP_PT * 8; wq += 512) {
            const int pp = wq >> 3, cg = wq & 7, ly = pp / SP_PT, lx = pp - ly * SP_PT;
            const int gy = py0 + ly, gx = px0 + lx;
            if (gy < p.PO && gx < p.PO) {
                half8 mx = *(const half8*)(stage + ((2 * ly) * SP_ST + 2 * lx) * SP_SS + cg * 16);
#pragma unroll
                for (int d = 1; d < 9; ++d) {
                    const half8 v = *(const half8*)(stage + ((2 * ly + d / 3) * SP_ST + 2 * lx + d % 3) * SP_SS + cg * 16);
                    mx = __builtin_elementwise_max(mx, v);
                }
                *(half8*)(p.pool + (long long)b * p.pool_img_stride + ((long long)gy * p.PO + gx) * 64 + cg * 8) = mx;
            }
        }
    }
}

hipError_t launch_stem_pool(const StemPoolParams& p, hipStream_t stream) {
    const long long total = (long long)p.n * p.tiles_y * p.tiles_x;
    if (total < 1 || (p.Wp & 1)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total < 1024 ? total : 1024);   // persistent: weight fragments are loaded once per workgroup
    hipLaunchKernelGGL(stem_pool_f16, dim3(grid), dim3(512), 0, stream, p);
    return hipGetLastError();
}

// Sums the split-K partial slabs in slice order, then the usual epilogue. One lane = 8 channels of
// one output row: 32-byte f32 reads per slice, one 16-byte f16 store. The slices' loads are issued eight (then four, two, one)
// at a time and added in slice order: as a plain loop over the run-time slice count the compiler waited for every slice's
// loads before it asked for the next one's, and the launch took k_slices L2 latencies (6.8 us on average at batch 1).
template <int NB>
__device__ __forceinline__ void splitk_add(const float*& src, long long slab, float v[8]) {
    f32x4 a[NB], b[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) { a[i] = *(const f32x4*)(src + i * slab); b[i] = *(const f32x4*)(src + i * slab + 4); }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[i][e]; v[4 + e] += b[i][e]; }
    src += NB * slab;
}
__global__ __launch_bounds__(256) void splitk_reduce_f16(const ConvParams p) {
    const int groups = p.cout8 >> 3;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.M * groups) return;
    const int m = (int)(t / groups), ch = (int)(t - (long long)m * groups) * 8;
    long long yo, ro;
    if (p.y_dense) { yo = (long long)m * p.ldy + ch; ro = (long long)m * p.ldres + ch; }
    else {
        const int PQ = p.P * p.Q, n = m / PQ, rem = m - n * PQ;
        yo = n * p.y_img_stride + (long long)rem * p.ldy + ch;
        ro = n * p.res_img_stride + (long long)rem * p.ldres + ch;
    }
    // (the residual row and the bias are asked for in front of the slabs: one latency for all of them)
    half8 rv;
    if (p.res) rv = load_residual<true>(p, m, ch, ro);
    const f32x4 b0 = *(const f32x4*)(p.bias + ch), b1 = *(const f32x4*)(p.bias + ch + 4);   // (bias is padded to coutPad)
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    {
        const float* src = p.partial + (long long)m * p.partial_ld + ch;
        const long long slab = (long long)p.M * p.partial_ld;
        int left = p.k_slices;
        for (; left >= 8; left -= 8) splitk_add<8>(src, slab, v);
        if (left & 4) splitk_add<4>(src, slab, v);
        if (left & 2) splitk_add<2>(src, slab, v);
        if (left & 1) splitk_add<1>(src, slab, v);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = v[e] + b0[e]; v[4 + e] = v[4 + e] + b1[e]; }
    if (p.res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)rv[e];
    }
    if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    if (ch + 8 > p.tanh_from) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (ch + e >= p.tanh_from) v[e] = spec_tanhf(v[e]);
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
    if (p.y) *(half8*)(p.y + yo) = o;
    if (p.y8) {
        const float* iv = p.y8_inv + ch;
        const unsigned lo = e4m3_pack4((float)o[0] * iv[0], (float)o[1] * iv[1], (float)o[2] * iv[2], (float)o[3] * iv[3]);
        const unsigned hi = e4m3_pack4((float)o[4] * iv[4], (float)o[5] * iv[5], (float)o[6] * iv[6], (float)o[7] * iv[7]);
        *(uint2*)(p.y8 + yo) = make_uint2(lo, hi);
    }
}

int conv_tile_ch(ConvTile t) {
    switch (t) { case TILE_128x128: case TILE_128x128_S3: case TILE_128x128_M16: case TILE_128x128_S3_M16: case TILE_128x128_K1: case TILE_128x128_FP8: case TILE_128x256: case TILE_128x256_M16: return 128; case TILE_64x256: case TILE_64x256_SMALLC: case TILE_64x256_K1: case TILE_64x64_S3: case TILE_64x64_FP8: return 64;
                 case TILE_32x256: return 32; case TILE_96x128_K1: return 96; case TILE_256x256: case TILE_256x256_M16: case TILE_256x256_FP8: return 256; }
    return 0;
}
int conv_tile_m(ConvTile t) {
    switch (t) { case TILE_128x128: case TILE_128x128_S3: case TILE_128x128_M16: case TILE_128x128_S3_M16: case TILE_128x128_K1: case TILE_128x128_FP8: case TILE_96x128_K1: return 128; case TILE_64x64_S3: case TILE_64x64_FP8: return 64; default: return 256; }
}
const char* conv_tile_symbol(ConvTile t) {
    switch (t) {
        case TILE_128x128: return "conv_igemm_f16<128,128,2,2,0,2>";
        case TILE_64x256: return "conv_igemm_f16<64,256,1,4,0,2>";
        case TILE_32x256: return "conv_igemm_f16<32,256,1,4,0,2>";
        case TILE_64x256_SMALLC: return "conv_igemm_f16<64,256,1,4,1,2>";
        case TILE_128x256: return "conv_igemm_f16<128,256,2,4,0,3>";
        case TILE_256x256: return "conv_igemm_f16<256,256,2,4,0,2>";
        case TILE_256x256_M16: return "conv_igemm_f16<256,256,2,4,0,2,mfma16>";
        case TILE_128x128_M16: return "conv_igemm_f16<128,128,2,2,0,2,mfma16>";
        case TILE_128x128_S3_M16: return "conv_igemm_f16<128,128,2,2,0,3,mfma16>";
        case TILE_128x128_S3: return "conv_igemm_f16<128,128,2,2,0,3>";
        case TILE_128x256_M16: return "conv_igemm_f16<128,256,2,4,0,2,mfma16>";
        case TILE_64x64_S3: return "conv_igemm_f16<64,64,2,2,0,3>";
        case TILE_128x128_K1: return "conv_igemm_f16<128,128,2,2,0,1>";
        case TILE_96x128_K1: return "conv_igemm_f16<96,128,2,2,0,1,mfma16>";
        case TILE_64x256_K1: return "conv_igemm_f16<64,256,1,4,0,1>";
        case TILE_256x256_FP8: return "conv_igemm_fp8<256,256,2,4>";
        case TILE_128x128_FP8: return "conv_igemm_fp8<128,128,2,2>";
        case TILE_64x64_FP8: return "conv_igemm_fp8<64,64,2,2>";
    }
    return "?";
}

hipError_t launch_splitk_reduce(const ConvParams& p, hipStream_t stream) {
    const long long work = (long long)p.M * (p.cout8 >> 3);
    hipLaunchKernelGGL(splitk_reduce_f16, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// One kernel launch (for split-K: the main kernel only).
hipError_t launch_conv(const ConvParams& p, ConvTile tile, hipStream_t stream) {
    const int tm = conv_tile_m(tile);
    const int n_m_tiles = (p.M + tm - 1) / tm - p.m_tile0;
    if (n_m_tiles < 1) return hipErrorInvalidValue;
    if (p.nlev > 0) {   // multi-level input (the shared prediction head over the whole pyramid): the tiles a head conv can get
        if (p.stride != 1 || p.nlev > 5) return hipErrorInvalidValue;
        if (p.k_slices > 1) {
            const dim3 gk((unsigned)(n_m_tiles * p.n_ch_tiles * p.k_slices));
            if (tile == TILE_128x128_S3) hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, true, 32, true>), gk, dim3(256), 0, stream, p);
            else if (tile == TILE_64x64_S3) hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, true, 32, true>), gk, dim3(256), 0, stream, p);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        const dim3 grid((unsigned)(n_m_tiles * p.n_ch_tiles));
        switch (tile) {
            case TILE_256x256_M16: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, true>), grid, dim3(512), 0, stream, p); break;
            case TILE_256x256_FP8:
                if (!p.scale) return hipErrorInvalidValue;
                hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, true, true>), grid, dim3(512), 0, stream, p);
                break;
            case TILE_128x128_FP8:
                if (!p.scale) return hipErrorInvalidValue;
                hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16, true, true>), grid, dim3(256), 0, stream, p);
                break;
            case TILE_128x256_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 256, 2, 4, false, 2, 2, false, 16, true>), grid, dim3(512), 0, stream, p); break;
            case TILE_128x256: hipLaunchKernelGGL((conv_igemm_f16<128, 256, 2, 4, false, 3, 1, false, 32, true>), grid, dim3(512), 0, stream, p); break;
            case TILE_128x128: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 32, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_K1: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 1, 2, false, 32, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_96x128_K1: hipLaunchKernelGGL((conv_igemm_f16<96, 128, 2, 2, false, 1, 2, false, 16, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 32, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 16, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_64x64_S3: hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, false, 32, true>), grid, dim3(256), 0, stream, p); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.w2) {   // fused 1x1 tail: single launches of the 256 x 256 tile whose one channel tile is the whole conv
        if (p.k_slices > 1 || p.n_ch_tiles != 1 || p.ch_tile0 || p.cout8 != 256 || p.res || p.x2 || p.nlev || !p.bias2 || !p.y2 || p.y || p.y8 || p.tanh_from < 256) return hipErrorInvalidValue;
        const dim3 grid((unsigned)(n_m_tiles * p.n_ch_tiles));
        if (tile == TILE_256x256_M16) hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, false, false, false, false, true>), grid, dim3(512), 0, stream, p);
        else if (tile == TILE_256x256_FP8 && p.scale) hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, false, true, false, false, true>), grid, dim3(512), 0, stream, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.x2) {   // two-source 1x1 form (a bottleneck block's last conv + its projection): the tiles dual_conv_tile() maps to
        if (p.R != 1 || p.S != 1 || p.pad != 0 || p.stride != 1 || p.res_up || p.k1steps < 1 || p.k1steps >= p.ksteps || p.C2 % 64 != 0) return hipErrorInvalidValue;
        if (p.k_slices > 1) {
            const dim3 gk((unsigned)(n_m_tiles * p.n_ch_tiles * p.k_slices));
            if (tile == TILE_128x128_S3) hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, true, 32, false, false, false, true>), gk, dim3(256), 0, stream, p);
            else if (tile == TILE_64x64_S3) hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, true, 32, false, false, false, true>), gk, dim3(256), 0, stream, p);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        const dim3 grid((unsigned)(n_m_tiles * p.n_ch_tiles));
        switch (tile) {
            case TILE_128x128: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 32, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_K1: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 1, 2, false, 32, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 32, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_64x64_S3: hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, false, 32, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 16, false, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_256x256_M16: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, false, false, false, true>), grid, dim3(512), 0, stream, p); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.k_slices > 1) {   // split-K main kernel; launch_splitk_reduce finishes it
        const dim3 gk((unsigned)(n_m_tiles * p.n_ch_tiles * p.k_slices));
        if (tile == TILE_128x128_S3) hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, true>), gk, dim3(256), 0, stream, p);
        else if (tile == TILE_64x64_S3) hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, true>), gk, dim3(256), 0, stream, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    const dim3 grid((unsigned)(n_m_tiles * p.n_ch_tiles));
    if (p.res_up) {   // the FPN lateral convs (1x1, 256 output channels): the tiles pick_tile / plan_conv can give them
        switch (tile) {
            case TILE_128x128: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 32, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_K1: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 1, 2, false, 32, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 32, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_64x64_S3: hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1, false, 32, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_128x128_S3_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 16, false, false, true>), grid, dim3(256), 0, stream, p); break;
            case TILE_256x256_M16: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, false, false, true>), grid, dim3(512), 0, stream, p); break;
            case TILE_256x256: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 32, false, false, true>), grid, dim3(512), 0, stream, p); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (tile) {
        case TILE_128x128: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_128x128_K1:
            if (p.R == 3 && p.S == 3) hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 1, 2, false, 32, false, false, false, false, false, true>), grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 1, 2>), grid, dim3(256), 0, stream, p);
            break;
        case TILE_64x256_K1: hipLaunchKernelGGL((conv_igemm_f16<64, 256, 1, 4, false, 1, 2>), grid, dim3(256), 0, stream, p); break;
        case TILE_64x256: hipLaunchKernelGGL((conv_igemm_f16<64, 256, 1, 4, false, 2, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_32x256: hipLaunchKernelGGL((conv_igemm_f16<32, 256, 1, 4, false, 2, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_64x256_SMALLC: hipLaunchKernelGGL((conv_igemm_f16<64, 256, 1, 4, true, 2, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_128x256: hipLaunchKernelGGL((conv_igemm_f16<128, 256, 2, 4, false, 3, 1>), grid, dim3(512), 0, stream, p); break;
        case TILE_128x128_S3: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_64x64_S3: hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 3, 1>), grid, dim3(256), 0, stream, p); break;
        case TILE_128x256_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 256, 2, 4, false, 2, 2, false, 16>), grid, dim3(512), 0, stream, p); break;
        case TILE_128x128_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16>), grid, dim3(256), 0, stream, p); break;
        case TILE_128x128_S3_M16: hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 3, 1, false, 16>), grid, dim3(256), 0, stream, p); break;
        case TILE_256x256_M16: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16>), grid, dim3(512), 0, stream, p); break;
        case TILE_256x256: hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2>), grid, dim3(512), 0, stream, p); break;
        case TILE_256x256_FP8:
            if (!p.scale) return hipErrorInvalidValue;
            hipLaunchKernelGGL((conv_igemm_f16<256, 256, 2, 4, false, 2, 2, false, 16, false, true>), grid, dim3(512), 0, stream, p);
            break;
        case TILE_128x128_FP8:   // launches with few 256 x 256 tiles (small batches): four times the workgroups, two per CU
            if (!p.scale) return hipErrorInvalidValue;
            hipLaunchKernelGGL((conv_igemm_f16<128, 128, 2, 2, false, 2, 1, false, 16, false, true>), grid, dim3(256), 0, stream, p);
            break;
        case TILE_64x64_FP8:     // ... and with at most two 128 x 128 tiles per CU: sixteen times the workgroups
            if (!p.scale) return hipErrorInvalidValue;
            hipLaunchKernelGGL((conv_igemm_f16<64, 64, 2, 2, false, 2, 1, false, 16, false, true>), grid, dim3(256), 0, stream, p);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace yh
