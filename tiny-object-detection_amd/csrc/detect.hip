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

__global__ __launch_bounds__(256) void det_softmax_cand(const DetectParams p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.n * p.P) return;
    const int b = (int)(t / p.P), pr = (int)(t % p.P);
    const int cell = pr / 3, a = pr - cell * 3, C = p.C;
    const half_t* z = p.heads + ((long long)b * p.cells + cell) * p.ldh + 12 + a * C;
    float m = (float)z[0];
    for (int c = 1; c < C; ++c) { const float v = (float)z[c]; m = v > m ? v : m; }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s = __fadd_rn(s, spec_expf(__fsub_rn((float)z[c], m)));
    for (int c = 1; c < C; ++c) {
        const float pc = __fdiv_rn(spec_expf(__fsub_rn((float)z[c], m)), s);
        if (pc > p.conf_thresh) {
            const int list = b * (C - 1) + (c - 1);
            const int slot = atomicAdd(&p.cls_count[list], 1);
            p.cand[(long long)list * p.P + slot] = make_uint2(__float_as_uint(pc), (unsigned)pr);
        }
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

#define YH_TOPK_MAX 256

__global__ __launch_bounds__(256) void det_class_nms(const DetectParams p) {
    __shared__ float sel_score[YH_TOPK_MAX];
    __shared__ int sel_prior[YH_TOPK_MAX];
    __shared__ float4 sel_box[YH_TOPK_MAX];
    const int list = blockIdx.x;  // b*(C-1) + c
    const int b = list / (p.C - 1);
    const int tid = threadIdx.x;
    const int nc = p.cls_count[list];
    const uint2* cand = p.cand + (long long)list * p.P;
    const int K = nc < p.top_k ? nc : p.top_k;
    // rank-by-counting: keys are unique (prior breaks score ties), so ranks are a permutation
    for (int i = tid; i < nc; i += 256) {
        const uint2 me = cand[i];
        const float si = __uint_as_float(me.x);
        int rank = 0;
        for (int j = 0; j < nc; ++j) {
            const uint2 o = cand[j];  // uniform address: one broadcast load per wave
            const float sj = __uint_as_float(o.x);
            rank += (sj > si || (sj == si && o.y < me.y)) ? 1 : 0;
        }
        if (rank < p.top_k) { sel_score[rank] = si; sel_prior[rank] = (int)me.y; }
    }
    __syncthreads();
    if (tid < K) {
        const int pr = sel_prior[tid], cell = pr / 3, a = pr - cell * 3;
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
    const long long so = (long long)list * p.top_k;
    for (int j = tid; j < p.top_k; j += 256) {
        float out = -1.0f;
        if (j < K) {
            const float4 bj = sel_box[j];
            bool keep = true;
            for (int i = 0; i < j; ++i) keep = keep && !(box_iou(sel_box[i], bj) > p.nms_thresh);
            if (keep) {
                out = sel_score[j];
                p.surv_prior[so + j] = sel_prior[j];
                *(float4*)(p.surv_box + (so + j) * 4) = bj;
            }
        }
        p.surv_score[so + j] = out;
    }
}

#define YH_SLOTS_MAX 16384

__global__ __launch_bounds__(1024) void det_frame_top(const DetectParams p) {
    __shared__ float vs[YH_SLOTS_MAX];
    __shared__ int vi[YH_SLOTS_MAX];
    __shared__ int V;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NS = (p.C - 1) * p.top_k;
    if (tid == 0) V = 0;
    __syncthreads();
    const float* ss = p.surv_score + (long long)b * NS;
    for (int i = tid; i < NS; i += 1024) {
        const float s = ss[i];
        if (s >= 0.0f) { const int pos = atomicAdd(&V, 1); vs[pos] = s; vi[pos] = i; }
    }
    __syncthreads();
    const int nv = V;
    for (int e = tid; e < nv; e += 1024) {
        const float se = vs[e];
        const int ie = vi[e];
        int rank = 0;
        for (int f = 0; f < nv; ++f) {
            const float sf = vs[f];
            rank += (sf > se || (sf == se && vi[f] < ie)) ? 1 : 0;
        }
        if (rank < p.max_dets) {
            const long long slot = (long long)b * NS + ie;
            yh_detection d;
            d.class_id = ie / p.top_k;
            d.prior = p.surv_prior[slot];
            d.score = se;
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
        }
    }
    if (tid == 0) p.det_count[b] = nv < p.max_dets ? nv : p.max_dets;
}

#define YH_DETS_MAX 128

__global__ __launch_bounds__(256) void det_masks(const DetectParams p) {
    __shared__ float coef[YH_DETS_MAX * 32];
    __shared__ float4 crop[YH_DETS_MAX];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nd = p.det_count[b];
    for (int i = tid; i < nd * 32; i += 256) {
        const int d = i >> 5, k = i & 31;
        const int pr = p.dets[(long long)b * p.max_dets + d].prior, cell = pr / 3, a = pr - cell * 3;
        coef[i] = (float)p.heads[((long long)b * p.cells + cell) * p.ldh + 12 + 3 * p.C + a * 32 + k];
    }
    for (int d = tid; d < nd; d += 256) crop[d] = *(const float4*)(p.det_crop + ((long long)b * p.max_dets + d) * 4);
    __syncthreads();
    const int npx = p.hp * p.wp;
    const int px = blockIdx.x * 256 + tid;
    if (px >= npx) return;
    const int y = px / p.wp, x = px - y * p.wp;
    const float fx = (float)x, fy = (float)y;
    float pv[32];
    const half_t* pp = p.proto + ((long long)b * npx + px) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const half8 v = *(const half8*)(pp + q * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[q * 8 + e] = (float)v[e];
    }
    uint8_t* mo = p.masks + (long long)b * p.max_dets * npx + px;
    for (int d = 0; d < nd; ++d) {
        const float* co = coef + d * 32;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __fmaf_rn(pv[k], co[k], acc);
        const float4 c = crop[d];
        const bool inside = fx >= c.x && fx < c.y && fy >= c.z && fy < c.w;
        mo[(long long)d * npx] = (uint8_t)((inside && acc > 0.0f) ? 1 : 0);
    }
}

int detect_launch_count() { return 5; }

const char* detect_stage_name(int stage) {
    static const char* n[5] = { "hipMemsetAsync:det_counts", "det_softmax_cand:tail", "det_class_nms:tail",
                                "det_frame_top:tail", "det_masks:tail" };
    return stage >= 0 && stage < 5 ? n[stage] : "?";
}

hipError_t launch_detect_stage(const DetectParams& p, int stage, hipStream_t s) {
    switch (stage) {
        case 0: return hipMemsetAsync(p.cls_count, 0, sizeof(int) * (size_t)p.n * (p.C - 1), s);
        case 1: {
            const long long np = (long long)p.n * p.P;
            hipLaunchKernelGGL(det_softmax_cand, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, p);
            break;
        }
        case 2: hipLaunchKernelGGL(det_class_nms, dim3((unsigned)(p.n * (p.C - 1))), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(det_frame_top, dim3((unsigned)p.n), dim3(1024), 0, s, p); break;
        case 4: hipLaunchKernelGGL(det_masks, dim3((unsigned)((p.hp * p.wp + 255) / 256), (unsigned)p.n), dim3(256), 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_detect(const DetectParams& p, hipStream_t s) {
    for (int st = 0; st < 5; ++st) {
        hipError_t e = launch_detect_stage(p, st, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace yh
