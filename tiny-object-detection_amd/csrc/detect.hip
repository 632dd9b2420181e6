// detect.hip — YOLACT detection tail as wavefront-primitive kernels.
//
// The reference does not implement this stage (/root/reference/src/yolact.rs:3-5, :93-95); it is
// the published YOLACT Detect + mask assembly as frozen in DESIGN.md §Spec-tail. Every float step
// uses an explicit round-to-nearest intrinsic in the spec's order (exp is the spec polynomial),
// so on equal head outputs the result is bit-identical to oracle/orc_detect.c.
//
//   K1 det_softmax_cand : one lane per prior; softmax over C logits (three passes over the 2*C
//                         bytes, no per-lane array); every (class, prior) with p > conf_thresh is
//                         appended to that class's candidate list (one atomic per candidate).
//   K2 det_class_nms    : one workgroup per (frame, class); rank-by-counting gives the top_k
//                         candidates in (score desc, prior asc) order without a sort network;
//                         SSD decode; Fast-NMS = upper-triangular IoU max test, lane j vs i < j.
//   K3 det_frame_top    : one workgroup per frame; compacts survivors into LDS, rank-by-counting
//                         in (score desc, class asc, rank asc) order, keeps max_dets, writes the
//                         detections and their crop windows.
//   K4 det_masks        : one lane per prototype pixel: the 32 prototype values stay in
//                         registers, coefficients of all detections sit in LDS; mask bit =
//                         (fma chain > 0) && inside crop window   [sigmoid(x) > 0.5 <=> x > 0].
#include "yh_internal.h"

namespace yh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define YH_DETS_MAX_K3 128

// K1: one workgroup = 64 consecutive head rows (cells) = 192 priors, one lane per prior.
// Rows are fetched whole with 16-byte coalesced loads and their conf part scattered to LDS as f32
// at [prior_local*C + c] (odd stride: conflict-free lane-per-prior reads). Pass 1 max, pass 2
// exp + sequential sum (exp values parked in LDS), pass 3 divide + threshold + append.
#define YH_K1_ROWS 64
__global__ __launch_bounds__(192) void det_softmax_cand(const DetectParams p) {
    extern __shared__ __attribute__((aligned(16))) float zs[];  // [192][C]
    const int C = p.C, tid = threadIdx.x, b = blockIdx.y;
    const int c0 = blockIdx.x * YH_K1_ROWS;       // first cell of this workgroup, within frame b
    const half_t* rows = p.heads + ((long long)b * p.cells + c0) * p.ldh;
    const int nrows = p.cells - c0 < YH_K1_ROWS ? p.cells - c0 : YH_K1_ROWS;
    const int chunks = p.ldh / 8;  // 16-byte chunks per row
    for (int i = tid; i < nrows * chunks; i += 192) {
        const int rl = i / chunks, j = i - rl * chunks;
        const half8 v = *(const half8*)(rows + (long long)rl * p.ldh + j * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = j * 8 + e - 12;
            if (q >= 0 && q < 3 * C) zs[rl * 3 * C + q] = (float)v[e];
        }
    }
    __syncthreads();
    const bool valid = tid / 3 < nrows;
    const int pr = (c0 + tid / 3) * 3 + tid % 3;
    float* z = zs + tid * C;
    float s = 1.0f;
    if (valid) {
        float m = z[0];
        for (int c = 1; c < C; ++c) { const float v = z[c]; m = v > m ? v : m; }
        s = 0.0f;
        for (int c = 0; c < C; ++c) { const float e = spec_expf(__fsub_rn(z[c], m)); z[c] = e; s = __fadd_rn(s, e); }
    }
    // wave-aggregated append: one atomic per (wave, class) that has any candidate
    const int lane = tid & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c = 1; c < C; ++c) {
        const float pc = valid ? __fdiv_rn(z[c], s) : 0.0f;
        const bool hit = pc > p.conf_thresh;
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;  // wave-uniform
        const int list = b * (C - 1) + (c - 1);
        int base = 0;
        if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&p.cls_count[list], __popcll(mask));
        base = __shfl(base, __ffsll((long long)mask) - 1);
        // (a list holds at most P entries - every prior once. det_class_nms leaves each counter at zero for the next step; should a step
        // ever be abandoned between this kernel and that one, the stale count must not carry the store past the list: ADVICE r3)
        const int slot = base + __popcll(mask & lt);
        if (hit && slot < p.P) p.cand[(long long)list * p.P + slot] = make_uint2(__float_as_uint(pc), (unsigned)pr);
    }
}

// K1 for a compile-time class count (the 81-class configuration). The conf part of 64 head rows is copied to LDS AS IT LIES
// IN THE ROW (dwords 6 .. 127 = halves 12 .. 255, coalesced dword loads, conflict-free dword stores; the row pitch of 123 dwords
// is odd), wave k of the three takes anchor k and lane l the row l: its 81 logits are halves 81 k .. 81 k + 80 of the row image,
// read one ds_read_u16 each (123 l + const: every lane its own bank). Same arithmetic, same order, same candidates as the
// generic kernel above; the candidates' order inside a class list is free (K2 ranks by (score, prior)).
// The reads are `volatile` on purpose: left to merge them the compiler builds wide ds_read_b128 / ds_read2_b32, and that form
// together with its packed-f32 exponentials gave scores 1e-7 .. 1e-3 off in lanes 48-63 of a wave under the graph-replayed
// two-stream step (tools/study/tail_vs_oracle_repeat.py; the guard is
// tests/test_gpu_fullsize.py::test_tail_exact_and_heads_repeatable_after_idle_gaps_on_the_two_stream_graph).
#define YH_K1_RS 123   // dwords per row image in LDS (122 used)
template <int C>
__global__ __launch_bounds__(192) void det_softmax_cand_c(const DetectParams p) {
    static_assert((12 + 3 * C + 1) / 2 - 6 <= YH_K1_RS - 1 && C % 2 == 1, "row image: dwords 6 .. (12 + 3 C) / 2");
    constexpr int ND = (12 + 3 * C + 1) / 2 - 6;   // 122 dwords of a row hold its 3 C logits (and one half of the next field)
    __shared__ uint32_t zw[YH_K1_ROWS * YH_K1_RS];
    const int tid = threadIdx.x, b = blockIdx.y, lane = tid & 63, k = tid >> 6;
    const int c0 = blockIdx.x * YH_K1_ROWS;
    const half_t* rows = p.heads + ((long long)b * p.cells + c0) * p.ldh;
    const int nrows = p.cells - c0 < YH_K1_ROWS ? p.cells - c0 : YH_K1_ROWS;
    // wave k copies rows k, k + 3, ...: four rows' loads in flight, then their stores
    for (int r0 = k; r0 < nrows; r0 += 12) {
        uint32_t v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rl = r0 + 3 * u;
            const uint32_t* src = (const uint32_t*)(rows + (long long)(rl < nrows ? rl : r0) * p.ldh) + 6;
            v0[u] = src[lane];
            v1[u] = src[lane < ND - 64 ? 64 + lane : 64];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rl = r0 + 3 * u;
            if (rl < nrows) {
                zw[rl * YH_K1_RS + lane] = v0[u];
                if (lane < ND - 64) zw[rl * YH_K1_RS + 64 + lane] = v1[u];
            }
        }
    }
    __syncthreads();
    const bool valid = lane < nrows;
    const int pr = (c0 + lane) * 3 + k;
    typedef const volatile __attribute__((address_space(3))) half_t lds_logit_t;   // (a plain volatile pointer became 81 flat loads)
    lds_logit_t* z = (lds_logit_t*)zw + lane * (2 * YH_K1_RS) + C * k;
    float e[C];
    float s = 1.0f;
    if (valid) {
#pragma unroll
        for (int c = 0; c < C; ++c) e[c] = (float)z[c];
        float m = e[0];
#pragma unroll
        for (int c = 1; c < C; ++c) m = e[c] > m ? e[c] : m;
        s = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) { e[c] = spec_expf(__fsub_rn(e[c], m)); s = __fadd_rn(s, e[c]); }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) e[c] = 0.0f;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // (rounding is monotone and conf_thresh is a float: e / s <= conf_thresh in real numbers => fl(e / s) <= conf_thresh. `lim` sits
    // 2^-10 below fl(conf_thresh * s), so e < lim proves "no candidate" without the division; every other case takes the exact test)
    const float lim = valid ? __fmul_rn(__fmul_rn(p.conf_thresh, s), 0.9990234375f) : 3.0e38f;
#pragma unroll
    for (int c = 1; c < C; ++c) {
        if (__ballot(e[c] >= lim) == 0ull) continue;  // wave-uniform: no lane can pass
        const float pc = valid ? __fdiv_rn(e[c], s) : 0.0f;
        const bool hit = pc > p.conf_thresh;
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;  // wave-uniform
        const int list = b * (C - 1) + (c - 1);
        int base = 0;
        if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&p.cls_count[list], __popcll(mask));
        base = __shfl(base, __ffsll((long long)mask) - 1);
        // (a list holds at most P entries - every prior once. det_class_nms leaves each counter at zero for the next step; should a step
        // ever be abandoned between this kernel and that one, the stale count must not carry the store past the list: ADVICE r3)
        const int slot = base + __popcll(mask & lt);
        if (hit && slot < p.P) p.cand[(long long)list * p.P + slot] = make_uint2(__float_as_uint(pc), (unsigned)pr);
    }
}

__device__ __forceinline__ float box_iou(const float4 a, const float4 b) {
    float iw = __fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x));
    float ih = __fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y));
    iw = iw < 0.0f ? 0.0f : iw;
    ih = ih < 0.0f ? 0.0f : ih;
    const float inter = __fmul_rn(iw, ih);
    const float aa = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const float ab = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    const float uni = __fsub_rn(__fadd_rn(aa, ab), inter);
    return uni > 0.0f ? __fdiv_rn(inter, uni) : 0.0f;
}

// Exact k-th largest of n UNIQUE 64-bit keys by MSB-first radix select: 8 passes of a 256-bin LDS
// histogram over the keys that still match the prefix; the bin holding the k-th key is found by one
// wave (4 bins per lane + a suffix scan by shuffles). Every thread of the workgroup must call it.
template <int NT, class KeyFn>
__device__ unsigned long long radix_select_kth(KeyFn key_of, int n, int k, int* hist, int* sh) {
    const int tid = threadIdx.x;
    unsigned long long prefix = 0, mask = 0;
    int kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += NT) hist[i] = 0;
        __syncthreads();
        // run-length aggregation per lane: the high bytes of score keys (sign, exponent) take two or three
        // values, and one LDS atomic per key on the same bin serialises the whole pass
        int run_bin = -1, run_cnt = 0;
        for (int i = tid; i < n; i += NT) {
            const unsigned long long key = key_of(i);
            if ((key & mask) == prefix) {
                const int bin = (int)((key >> shift) & 255ull);
                if (bin == run_bin) ++run_cnt;
                else { if (run_cnt) atomicAdd(&hist[run_bin], run_cnt); run_bin = bin; run_cnt = 1; }
            }
        }
        if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
        __syncthreads();
        if (tid < 64) {
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int mine = h0 + h1 + h2 + h3;
            int incl = mine;  // inclusive suffix sum over lanes >= tid
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_down(incl, d);
                incl += (tid + d < 64) ? o : 0;
            }
            const int above = incl - mine;
            if (above < kk && kk <= incl) {  // exactly one lane
                int c = above, bsel;
                if (c + h3 >= kk) bsel = 3;
                else { c += h3; if (c + h2 >= kk) bsel = 2; else { c += h2; if (c + h1 >= kk) bsel = 1; else { c += h1; bsel = 0; } } }
                sh[0] = 4 * tid + bsel;
                sh[1] = kk - c;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)(unsigned)sh[0] << shift;
        mask |= 0xFFull << shift;
        kk = sh[1];
        __syncthreads();
    }
    return prefix;
}

#define YH_TOPK_MAX 256
#define YH_K2_STAGE 4096

// K2: one workgroup per (frame, class).
__global__ __launch_bounds__(256) void det_class_nms(const DetectParams p) {
    __shared__ unsigned long long stage[YH_K2_STAGE];   // candidate keys (when they fit)
    __shared__ unsigned long long sel[YH_TOPK_MAX];     // the K selected keys, then sorted
    __shared__ unsigned long long sorted[YH_TOPK_MAX];
    __shared__ float4 sel_box[YH_TOPK_MAX];
    __shared__ int supp[YH_TOPK_MAX];
    __shared__ int hist[256];
    __shared__ int sh[4];
    const int list = blockIdx.x;  // b*(C-1) + c
    const int b = list / (p.C - 1);
    const int tid = threadIdx.x;
    // This workgroup is the one consumer of its list's candidate counter: it takes the count and leaves the counter zero for
    // the next step, so that no memset node is needed in front of K1 (a captured step then consists of kernel nodes only:
    // profiles/r03_fault_audit.md). yh_create zeroes the counters once.
    __shared__ int sh_nc;
    if (tid == 0) { sh_nc = p.cls_count[list]; p.cls_count[list] = 0; }
    __syncthreads();
    const int nc = sh_nc;
    const uint2* cand = p.cand + (long long)list * p.P;
    const int K = nc < p.top_k ? nc : p.top_k;
    const long long so = (long long)list * p.top_k;
    if (nc == 0) {  // uniform
        for (int j = tid; j < p.top_k; j += 256) p.surv_score[so + j] = -1.0f;
        return;
    }
    // key = score bits (positive float: monotone) : ~prior  -> larger key = better, keys unique
    const bool staged = nc <= YH_K2_STAGE;
    if (staged) {
        for (int i = tid; i < nc; i += 256) { const uint2 c = cand[i]; stage[i] = ((unsigned long long)c.x << 32) | (unsigned long long)(0xFFFFFFFFu - c.y); }
    }
    if (tid == 0) sh[2] = 0;
    __syncthreads();
    auto key_of = [&](int i) -> unsigned long long {
        if (staged) return stage[i];
        const uint2 c = cand[i];
        return ((unsigned long long)c.x << 32) | (unsigned long long)(0xFFFFFFFFu - c.y);
    };
    unsigned long long T = 0;
    if (nc > p.top_k) T = radix_select_kth<256>(key_of, nc, K, hist, sh);
    for (int i = tid; i < nc; i += 256) {
        const unsigned long long key = key_of(i);
        if (key >= T) sel[atomicAdd(&sh[2], 1)] = key;  // exactly K keys pass
    }
    __syncthreads();
    if (tid < K) {  // order the K selected keys by counting
        const unsigned long long me = sel[tid];
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += sel[j] > me ? 1 : 0;
        sorted[rank] = me;
    }
    __syncthreads();
    if (tid < K) {
        const int pr = (int)(0xFFFFFFFFu - (unsigned)(sorted[tid] & 0xFFFFFFFFull)), cell = pr / 3, a = pr - cell * 3;
        const half_t* l = p.heads + ((long long)b * p.cells + cell) * p.ldh + a * 4;
        const float4 q = *(const float4*)(p.priors + (long long)pr * 4);
        const float l0 = (float)l[0], l1 = (float)l[1], l2 = (float)l[2], l3 = (float)l[3];
        const float cx = __fadd_rn(q.x, __fmul_rn(__fmul_rn(l0, 0.1f), q.z));
        const float cy = __fadd_rn(q.y, __fmul_rn(__fmul_rn(l1, 0.1f), q.w));
        const float w = __fmul_rn(q.z, spec_expf(__fmul_rn(l2, 0.2f)));
        const float h = __fmul_rn(q.w, spec_expf(__fmul_rn(l3, 0.2f)));
        float4 bx;
        bx.x = __fsub_rn(cx, __fmul_rn(w, 0.5f));
        bx.y = __fsub_rn(cy, __fmul_rn(h, 0.5f));
        bx.z = __fadd_rn(w, bx.x);
        bx.w = __fadd_rn(h, bx.y);
        sel_box[tid] = bx;
    }
    __syncthreads();
    // Fast-NMS: box j survives iff no better-ranked box i < j overlaps it by more than the threshold.
    // The K (K - 1) / 2 pairs are dealt evenly over the workgroup (row j has j pairs: rows j and K - j
    // are folded into one row of K), each hit sets a flag; a lane per row would make the last lane walk
    // K - 1 boxes while the first walks none (30-60 us of the kernel at K = 200).
    for (int j = tid; j < YH_TOPK_MAX; j += 256) supp[j] = 0;
    __syncthreads();
    {
        const int half_rows = (K - 1) / 2, folded = half_rows * K;
        for (int t = tid; t < folded; t += 256) {
            const int r = t / K, c = t - r * K, j0 = r + 1;
            const int j = c < j0 ? j0 : K - j0, i = c < j0 ? c : c - j0;
            if (box_iou(sel_box[i], sel_box[j]) > p.nms_thresh) supp[j] = 1;
        }
        if ((K & 1) == 0 && K >= 2) {   // the unpaired middle row
            const int j = K / 2;
            for (int i = tid; i < j; i += 256)
                if (box_iou(sel_box[i], sel_box[j]) > p.nms_thresh) supp[j] = 1;
        }
    }
    __syncthreads();
    for (int j = tid; j < p.top_k; j += 256) {
        float out = -1.0f;
        if (j < K) {
            const float4 bj = sel_box[j];
            const bool keep = supp[j] == 0;
            if (keep) {
                out = __uint_as_float((unsigned)(sorted[j] >> 32));
                p.surv_prior[so + j] = (int)(0xFFFFFFFFu - (unsigned)(sorted[j] & 0xFFFFFFFFull));
                *(float4*)(p.surv_box + (so + j) * 4) = bj;
            }
        }
        p.surv_score[so + j] = out;
    }
}

#define YH_SLOTS_MAX 16384

// K3: one workgroup per frame: survivors -> exact top max_dets in (score desc, slot asc) order.
__global__ __launch_bounds__(1024) void det_frame_top(const DetectParams p) {
    __shared__ unsigned long long keys[YH_SLOTS_MAX];
    __shared__ unsigned long long sel[YH_DETS_MAX_K3];
    __shared__ int hist[256];
    __shared__ int sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NS = (p.C - 1) * p.top_k;
    if (tid == 0) { sh[2] = 0; sh[3] = 0; }
    __syncthreads();
    const float* ss = p.surv_score + (long long)b * NS;
    for (int i = tid; i < NS; i += 1024) {
        const float s = ss[i];
        if (s >= 0.0f) keys[atomicAdd(&sh[2], 1)] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
    __syncthreads();
    const int nv = sh[2];
    const int K = nv < p.max_dets ? nv : p.max_dets;
    if (tid == 0) p.det_count[b] = K;
    if (nv == 0) return;
    unsigned long long T = 0;
    if (nv > p.max_dets) T = radix_select_kth<1024>([&](int i) { return keys[i]; }, nv, K, hist, sh);
    for (int i = tid; i < nv; i += 1024)
        if (keys[i] >= T) sel[atomicAdd(&sh[3], 1)] = keys[i];
    __syncthreads();
    if (tid < K) {
        const unsigned long long me = sel[tid];
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += sel[j] > me ? 1 : 0;
        const int ie = (int)(0xFFFFFFFFu - (unsigned)(me & 0xFFFFFFFFull));
        const long long slot = (long long)b * NS + ie;
        yh_detection d;
        d.class_id = ie / p.top_k;
        d.prior = p.surv_prior[slot];
        d.score = __uint_as_float((unsigned)(me >> 32));
        const float4 bx = *(const float4*)(p.surv_box + slot * 4);
        d.box[0] = bx.x; d.box[1] = bx.y; d.box[2] = bx.z; d.box[3] = bx.w;
        p.dets[(long long)b * p.max_dets + rank] = d;
        const float fw = (float)p.wp, fh = (float)p.hp;
        const float x1 = __fmul_rn(bx.x, fw), x2 = __fmul_rn(bx.z, fw);
        const float y1 = __fmul_rn(bx.y, fh), y2 = __fmul_rn(bx.w, fh);
        float xa = __fsub_rn(fminf(x1, x2), 1.0f), xb = __fadd_rn(fmaxf(x1, x2), 1.0f);
        float ya = __fsub_rn(fminf(y1, y2), 1.0f), yb = __fadd_rn(fmaxf(y1, y2), 1.0f);
        xa = xa < 0.0f ? 0.0f : xa;
        ya = ya < 0.0f ? 0.0f : ya;
        xb = xb > fw ? fw : xb;
        yb = yb > fh ? fh : yb;
        *(float4*)(p.det_crop + ((long long)b * p.max_dets + rank) * 4) = make_float4(xa, xb, ya, yb);
        // (round 5) ... and its 32 mask coefficients as f32, dense: the mask kernel's 75 workgroups per frame each gathered them again out of
        // the head rows - two dependent loads per coefficient, thirteen rounds per thread - before their first pixel
        const int cell = d.prior / 3, a = d.prior - cell * 3;
        const half_t* hc = p.heads + ((long long)b * p.cells + cell) * p.ldh + 12 + 3 * p.C + a * 32;
        float* co = p.det_coef + ((long long)b * p.max_dets + rank) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half_t v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = hc[q * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) co[q * 8 + e] = (float)v[e];
        }
    }
}

#define YH_DETS_MAX 128

__global__ __launch_bounds__(256) void det_masks(const DetectParams p) {
    __shared__ float coef[YH_DETS_MAX * 32];
    __shared__ float4 crop[YH_DETS_MAX];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nd = p.det_count[b];
    for (int i = tid; i < nd * 32; i += 256) coef[i] = p.det_coef[(long long)b * p.max_dets * 32 + i];   // (written by det_frame_top)
    // (round 5) which detections can touch this workgroup's pixels at all: its 256 pixels are rows y_lo .. y_hi of the prototype map, and a
    // detection whose crop window misses those rows is all zeros here - four such detections cost a wave one dword store, not four crop
    // tests and ballots (on a noise frame three quarters of the (workgroup, detection) pairs: 0.082 -> ... ms at batch 64)
    __shared__ unsigned rel[YH_DETS_MAX / 32];
    if (tid < YH_DETS_MAX / 32) rel[tid] = 0u;
    __syncthreads();
    const int npx = p.hp * p.wp;
    {
        const int p_lo = blockIdx.x * 256, p_hi = min(p_lo + 255, npx - 1);
        const float y_lo = (float)(p_lo / p.wp), y_hi = (float)(p_hi / p.wp);
        for (int d = tid; d < nd; d += 256) {
            const float4 c = *(const float4*)(p.det_crop + ((long long)b * p.max_dets + d) * 4);
            crop[d] = c;
            if (y_hi >= c.z && y_lo < c.w) atomicOr(&rel[d >> 5], 1u << (d & 31));   // some row y in [y_lo, y_hi] has c.z <= y < c.w
        }
    }
    __syncthreads();
    const int px = blockIdx.x * 256 + tid;
    const bool live = px < npx;
    const int y = px / p.wp, x = px - y * p.wp;
    const float fx = (float)x, fy = (float)y;
    float pv[32];
    const half_t* pp = p.proto + ((long long)b * npx + (live ? px : 0)) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const half8 v = *(const half8*)(pp + q * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[q * 8 + e] = (float)v[e];
    }
    // Round 5: FOUR detections per store instruction. A lane decides one pixel of one detection; written a byte per lane that is a 64-byte
    // store per wave and detection - 1.9 M store instructions for the 122 MB of a batch-64 step, which is what the kernel waited for
    // (0.098 ms alone on the chip at the end of the step). The wave's 64 decisions of a detection are one ballot; lane (j, q) = (lane >> 4,
    // lane & 15) then writes pixels 4 q .. 4 q + 3 of detection d + j as ONE dword (bit i of a nibble -> byte i: a multiplication by
    // 0x00204081 and a mask), so a wave stores 4 x 64 bytes at once. Same bytes; hp * wp is a multiple of 4 (hp and wp are even: the
    // prototypes are a x2 upsample), so a dword is wholly inside the mask or wholly past its end.
    const int lane = tid & 63, lj = lane >> 4, lq = lane & 15;
    const int wave_px0 = px - lane;
    uint8_t* mo = p.masks + (long long)b * p.max_dets * npx + wave_px0 + 4 * lq;
    const bool dword_live = wave_px0 + 4 * lq < npx;
    // small batches: the detections are dealt over gridDim.z groups so that more than hp*wp/256
    // workgroups exist (one frame: 75 workgroups on 256 CUs otherwise)
    const int per = ((p.max_dets + (int)gridDim.z - 1) / (int)gridDim.z + 3) & ~3;   // (whole groups of four detections: a group is one store)
    const int d0 = (int)blockIdx.z * per, d1 = d0 + per < nd ? d0 + per : nd;
    for (int d = d0; d < d1; d += 4) {
        unsigned long long m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = 0ull;
            if (d + j < d1 && ((rel[(d + j) >> 5] >> ((d + j) & 31)) & 1u)) {   // (wave-uniform)
                const float4 c = crop[d + j];
                const bool inside = live && fx >= c.x && fx < c.y && fy >= c.z && fy < c.w;
                float acc = 0.0f;
                if (__ballot(inside) != 0ull) {   // wave-uniform: no lane of these 64 pixels lies in the crop window -> all zeros
                    const float* co = coef + (d + j) * 32;
#pragma unroll
                    for (int k = 0; k < 32; ++k) acc = __fmaf_rn(pv[k], co[k], acc);
                }
                m[j] = __ballot(inside && acc > 0.0f);
            }
        }
        const unsigned long long mine = lj == 0 ? m[0] : (lj == 1 ? m[1] : (lj == 2 ? m[2] : m[3]));
        const unsigned nib = (unsigned)(mine >> (4 * lq)) & 15u;
        if (dword_live && d + lj < d1) *(unsigned*)(mo + (long long)(d + lj) * npx) = (nib * 0x00204081u) & 0x01010101u;
    }
}

int detect_launch_count() { return 4; }

const char* detect_stage_name(int stage) {
    static const char* n[4] = { "det_softmax_cand:tail", "det_class_nms:tail", "det_frame_top:tail", "det_masks:tail" };
    return stage >= 0 && stage < 4 ? n[stage] : "?";
}

hipError_t launch_detect_stage(const DetectParams& p, int stage, hipStream_t s) {
    switch (stage) {
        case 0: {   // (the candidate counters are zero on entry: K2 re-zeroes what it consumes)
            const dim3 grid((unsigned)((p.cells + YH_K1_ROWS - 1) / YH_K1_ROWS), (unsigned)p.n);
            // (the row-image form reads dwords 6 .. 127 of a head row: rows of at least 256 halves, dword aligned - ldh = 352 at 81 classes)
            if (p.C == 81 && !p.k1_generic && p.ldh >= 256 && p.ldh % 2 == 0) hipLaunchKernelGGL(det_softmax_cand_c<81>, grid, dim3(192), 0, s, p);
            else hipLaunchKernelGGL(det_softmax_cand, grid, dim3(192), (size_t)192 * p.C * sizeof(float), s, p);
            break;
        }
        case 1: hipLaunchKernelGGL(det_class_nms, dim3((unsigned)(p.n * (p.C - 1))), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(det_frame_top, dim3((unsigned)p.n), dim3(1024), 0, s, p); break;
        case 3: hipLaunchKernelGGL(det_masks, dim3((unsigned)((p.hp * p.wp + 255) / 256), (unsigned)p.n, p.n <= 2 ? 8u : (p.n <= 8 ? 2u : 1u)), dim3(256), 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_detect(const DetectParams& p, hipStream_t s) {
    for (int st = 0; st < 4; ++st) {
        hipError_t e = launch_detect_stage(p, st, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace yh
