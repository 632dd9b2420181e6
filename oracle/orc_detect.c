/*
 * orc_detect.c — CPU restatement of the YOLACT detection tail. TEST INFRASTRUCTURE.
 * Compile with -ffp-contract=off.
 *
 * PARITY UNPINNED against the reference: the reference does not implement this stage
 * (/root/reference/src/yolact.rs:3-5, :93-95 — "Not enough time ... to complete the yolact
 * detection cleanup implementation"). Restates the published YOLACT Detect + postprocess
 * (softmax, conf threshold, per-class top-k, SSD decode with variances 0.1/0.2, Fast-NMS,
 * top max_dets, mask = sigmoid(proto . coeff) > 0.5 cropped to the box) as frozen in
 * DESIGN.md §Spec-tail. Every float step is a single IEEE operation in a fixed order (exp is the
 * spec polynomial below, not libm), so a conforming implementation is bit-exact on equal inputs.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* exp(x): Cody-Waite reduction by ln2 (hi/lo), degree-5 polynomial in fmaf Horner form, 2^n by
 * exponent-field add. Inputs are clamped to [-87, 88]. */
float orc_spec_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float n = rintf(t);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    int32_t bits;
    memcpy(&bits, &y, 4);
    bits += (int32_t)n << 23;
    memcpy(&y, &bits, 4);
    return y;
}

/* tanh(x) = sign(x) * (1 - e) / (1 + e), e = exp(-2|x|). */
float orc_spec_tanhf(float x) {
    float a = fabsf(x);
    float e = orc_spec_expf(-2.0f * a);
    float t = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -t : t;
}

typedef struct { float score; int prior; } cand;

static int cand_cmp(const void* a, const void* b) { /* score desc, prior asc */
    const cand* x = (const cand*)a; const cand* y = (const cand*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return x->prior - y->prior;
}

typedef struct { float score; int cls, rank, prior; float box[4]; } surv;

static int surv_cmp(const void* a, const void* b) { /* score desc, class asc, rank asc */
    const surv* x = (const surv*)a; const surv* y = (const surv*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    if (x->cls != y->cls) return x->cls - y->cls;
    return x->rank - y->rank;
}

static void decode(const float* l, const float* p, float* b) {
    float cx = p[0] + (l[0] * 0.1f) * p[2];
    float cy = p[1] + (l[1] * 0.1f) * p[3];
    float w = p[2] * orc_spec_expf(l[2] * 0.2f);
    float h = p[3] * orc_spec_expf(l[3] * 0.2f);
    b[0] = cx - w * 0.5f;
    b[1] = cy - h * 0.5f;
    b[2] = w + b[0];
    b[3] = h + b[1];
}

static float iou(const float* a, const float* b) {
    float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
    float ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
    if (iw < 0.0f) iw = 0.0f;
    if (ih < 0.0f) ih = 0.0f;
    float inter = iw * ih;
    float aa = (a[2] - a[0]) * (a[3] - a[1]);
    float ab = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = (aa + ab) - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

int orc_detect(const orc_det_cfg* cfg, const float* loc, const float* conf, const float* mask,
               const float* proto, const float* priors, int P, int hp, int wp,
               orc_detection* dets, uint8_t* masks) {
    int C = cfg->num_classes, K = cfg->top_k;
    float* prob = (float*)malloc((size_t)P * C * sizeof(float));
    float* e = (float*)malloc((size_t)C * sizeof(float));
    for (int p = 0; p < P; ++p) {
        const float* z = conf + (size_t)p * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) if (z[c] > m) m = z[c];
        float s = 0.0f;
        for (int c = 0; c < C; ++c) { e[c] = orc_spec_expf(z[c] - m); s = s + e[c]; }
        for (int c = 0; c < C; ++c) prob[(size_t)p * C + c] = e[c] / s;
    }
    free(e);
    cand* cl = (cand*)malloc((size_t)P * sizeof(cand));
    surv* sv = (surv*)malloc((size_t)(C - 1) * K * sizeof(surv));
    float* boxes = (float*)malloc((size_t)K * 4 * sizeof(float));
    int ns = 0;
    for (int c = 1; c < C; ++c) {
        int nc = 0;
        for (int p = 0; p < P; ++p) {
            float s = prob[(size_t)p * C + c];
            if (s > cfg->conf_thresh) { cl[nc].score = s; cl[nc].prior = p; ++nc; }
        }
        qsort(cl, (size_t)nc, sizeof(cand), cand_cmp);
        if (nc > K) nc = K;
        for (int j = 0; j < nc; ++j) decode(loc + (size_t)cl[j].prior * 4, priors + (size_t)cl[j].prior * 4, boxes + j * 4);
        for (int j = 0; j < nc; ++j) {
            int keep = 1;
            for (int i = 0; i < j; ++i)
                if (iou(boxes + i * 4, boxes + j * 4) > cfg->nms_thresh) { keep = 0; break; }
            if (keep) {
                surv* s = &sv[ns++];
                s->score = cl[j].score; s->cls = c - 1; s->rank = j; s->prior = cl[j].prior;
                memcpy(s->box, boxes + j * 4, 16);
            }
        }
    }
    qsort(sv, (size_t)ns, sizeof(surv), surv_cmp);
    int nd = ns < cfg->max_dets ? ns : cfg->max_dets;
    for (int d = 0; d < nd; ++d) {
        dets[d].class_id = sv[d].cls;
        dets[d].prior = sv[d].prior;
        dets[d].score = sv[d].score;
        memcpy(dets[d].box, sv[d].box, 16);
        if (!masks) continue;
        const float* co = mask + (size_t)sv[d].prior * 32;
        const float* b = sv[d].box;
        float x1 = b[0] * (float)wp, x2 = b[2] * (float)wp, y1 = b[1] * (float)hp, y2 = b[3] * (float)hp;
        float xa = fminf(x1, x2) - 1.0f, xb = fmaxf(x1, x2) + 1.0f;
        float ya = fminf(y1, y2) - 1.0f, yb = fmaxf(y1, y2) + 1.0f;
        if (xa < 0.0f) xa = 0.0f;
        if (ya < 0.0f) ya = 0.0f;
        if (xb > (float)wp) xb = (float)wp;
        if (yb > (float)hp) yb = (float)hp;
        uint8_t* mo = masks + (size_t)d * hp * wp;
        for (int y = 0; y < hp; ++y)
            for (int x = 0; x < wp; ++x) {
                const float* pp = proto + ((size_t)y * wp + x) * 32;
                float acc = 0.0f;
                for (int k = 0; k < 32; ++k) acc = fmaf(pp[k], co[k], acc);
                int inside = (float)x >= xa && (float)x < xb && (float)y >= ya && (float)y < yb;
                mo[(size_t)y * wp + x] = (uint8_t)(inside && acc > 0.0f); /* sigmoid(acc) > 0.5 */
            }
    }
    free(prob); free(cl); free(sv); free(boxes);
    return nd;
}
