/*
 * orc_ref.c — CPU restatement of the reference's own pre/post-processing. TEST INFRASTRUCTURE.
 * Compile with -ffp-contract=off: every float expression below is one IEEE operation per
 * operator, as the Rust source evaluates it (rustc never contracts a*b+c).
 *
 * Follows /root/reference/src/yolact.rs and src/scene.rs line by line (cited per function).
 * Release-build Rust semantics are assumed for integer overflow (wrapping); a debug build of the
 * reference would panic at yolact.rs:61/:69 instead.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* yolact.rs:134-140 and :195-201 — u32::to_be_bytes(px)[..3] */
void orc_unpack_rgb(const uint32_t* px, size_t n, uint8_t* rgb) {
    for (size_t i = 0; i < n; ++i) {
        rgb[3 * i + 0] = (uint8_t)(px[i] >> 24);
        rgb[3 * i + 1] = (uint8_t)(px[i] >> 16);
        rgb[3 * i + 2] = (uint8_t)(px[i] >> 8);
    }
}

/* yolact.rs:213-214, :230-231 — u32::from_be_bytes([c0,c1,c2,0]); same packing as scene.rs:86 */
void orc_pack_rgb(const uint8_t* rgb, size_t n, uint32_t* px) {
    for (size_t i = 0; i < n; ++i)
        px[i] = ((uint32_t)rgb[3 * i] << 24) | ((uint32_t)rgb[3 * i + 1] << 16) |
                ((uint32_t)rgb[3 * i + 2] << 8);
}

/* yolact.rs:177 — scale * (((x as i32) - zero_point) as f32) */
void orc_dequant_u8(const uint8_t* q, size_t n, float scale, int32_t zero_point, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = scale * (float)((int32_t)q[i] - zero_point);
}

/* yolact.rs:108-118 — running max starting at 0.0, strict '>', over the first four logits of each
 * chunk of `nch`; then the four-arm match on the resulting bool pattern. */
void orc_gated_argmax(const float* dets, int ncells, int nch, uint8_t* classes) {
    for (int cell = 0; cell < ncells; ++cell) {
        const float* chunk = dets + (size_t)cell * nch;
        float max = 0.0f;
        int cls[4];
        for (int i = 0; i < 4; ++i) {
            if (chunk[i] > max) { max = chunk[i]; cls[i] = 1; } else cls[i] = 0;
        }
        uint8_t out;
        if (!cls[0] && cls[1] && !cls[2] && !cls[3]) out = 1;        /* [false,true,false,false] */
        else if (!cls[0] && cls[2] && !cls[3]) out = 2;              /* [false,_,true,false]      */
        else if (!cls[0] && cls[3]) out = 3;                         /* [false,_,_,true]          */
        else out = 0;                                                /* everything else           */
        classes[cell] = out;
    }
}

/* yolact.rs:52-88 — terrible_id + nested flood_fill, simulated literally with a pop budget.
 * `px - 1` and `px - 28` on usize wrap in release builds and then fail `img.get`. */
int orc_terrible_id(const uint8_t* img, int ncells, int grid_w, int8_t* out) {
    const size_t N = (size_t)ncells;
    const size_t budget = 4u * 1000u * 1000u;
    size_t cap = 1024, pops = 0;
    size_t* set = (size_t*)malloc(cap * sizeof(size_t));
    int8_t id = -1;
    for (size_t i = 0; i < N; ++i) out[i] = -1;
    for (size_t start = 0; start < N; ++start) {
        if (!(img[start] == 3 && out[start] == -1)) continue;
        id = (int8_t)(uint8_t)((uint8_t)id + 1u); /* i8 wraps past 127 in release */
        size_t len = 0;
        set[len++] = start;
        while (len > 0) {
            size_t px = set[--len];
            if (++pops > budget) { free(set); return 1; }
            size_t nb[4] = { px - 1, px + 1, px - (size_t)grid_w, px + (size_t)grid_w };
            for (int k = 0; k < 4; ++k) {
                size_t q = nb[k];
                if (q < N && img[q] == 3) { /* img.get(q) == Some(&3) */
                    out[q] = id;
                    if (len + 1 >= cap) { cap *= 2; set = (size_t*)realloc(set, cap * sizeof(size_t)); }
                    set[len++] = q;
                }
            }
        }
    }
    free(set);
    return 0;
}

/* SANE mode: 4-connected components of class-3 cells, no row wrap, ids 0,1,2.. in raster order of
 * each component's first cell, saturating at 127. Not reference behaviour. */
void orc_sane_id(const uint8_t* img, int gh, int gw, int8_t* out) {
    int n = gh * gw, id = -1;
    int* stack = (int*)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) out[i] = -1;
    for (int s = 0; s < n; ++s) {
        if (img[s] != 3 || out[s] != -1) continue;
        if (id < 127) ++id;
        int len = 0;
        stack[len++] = s;
        out[s] = (int8_t)id;
        while (len) {
            int p = stack[--len], y = p / gw, x = p % gw;
            int qs[4] = { x > 0 ? p - 1 : -1, x < gw - 1 ? p + 1 : -1, y > 0 ? p - gw : -1, y < gh - 1 ? p + gw : -1 };
            for (int k = 0; k < 4; ++k) {
                int q = qs[k];
                if (q >= 0 && img[q] == 3 && out[q] == -1) { out[q] = (int8_t)id; stack[len++] = q; }
            }
        }
    }
    free(stack);
}

/* yolact.rs:90-131 — postprocess: gated argmax, ids, pack, x8 nearest upsample. */
int orc_postprocess_tile(const float* cells, int grid, int nch, int mode, uint32_t* out) {
    int ncells = grid * grid, S = grid * 8;
    uint8_t* classes = (uint8_t*)malloc((size_t)ncells);
    int8_t* ids = (int8_t*)malloc((size_t)ncells);
    orc_gated_argmax(cells, ncells, nch, classes);
    int diverged = 0;
    if (mode == 0) diverged = orc_terrible_id(classes, ncells, grid, ids);
    else orc_sane_id(classes, grid, grid, ids);
    if (!diverged) {
        for (int r = 0; r < grid; ++r)
            for (int c = 0; c < grid; ++c) {
                uint32_t cls = classes[r * grid + c];
                uint32_t idu = (uint32_t)(int32_t)ids[r * grid + c]; /* `id as u32`: sign-extends */
                /* yolact.rs:127: `(cls as u32) << 24 & (id as u32) << 16` — '&', not '|' (A8') */
                uint32_t v = mode == 0 ? ((cls << 24) & (idu << 16))
                                       : ((cls << 24) | ((idu & 0xFFu) << 16));
                for (int dy = 0; dy < 8; ++dy)
                    for (int dx = 0; dx < 8; ++dx)
                        out[(size_t)(r * 8 + dy) * S + c * 8 + dx] = v;
            }
    }
    free(classes);
    free(ids);
    return diverged;
}

/* image 0.24.1 imageops::sample — Triangle kernel, support 1.0 (PARITY UNPINNED: crate not vendored). */
static float triangle_kernel(float x) {
    float a = fabsf(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}

static long clamp_l(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Weights of one output coordinate: returns left, writes n weights (normalised). */
static int sample_weights(int out_i, int in_size, int out_size, float* ws, int* n_out) {
    float ratio = (float)in_size / (float)out_size;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = 1.0f * sratio;
    float inputc = ((float)out_i + 0.5f) * ratio;
    long left = (long)floorf(inputc - src_support);
    left = clamp_l(left, 0, (long)in_size - 1);
    long right = (long)ceilf(inputc + src_support);
    right = clamp_l(right, left + 1, (long)in_size);
    inputc = inputc - 0.5f;
    float sum = 0.0f;
    int n = 0;
    for (long i = left; i < right; ++i) {
        float w = triangle_kernel(((float)i - inputc) / sratio);
        ws[n++] = w;
        sum += w;
    }
    for (int i = 0; i < n; ++i) ws[i] /= sum;
    *n_out = n;
    return (int)left;
}

void orc_resize_triangle_rgb8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    if (sw == dw && sh == dh) { memcpy(dst, src, (size_t)sw * sh * 3); return; }
    /* vertical pass into an unclamped f32 image (Rgba32FImage in 0.24), then horizontal pass */
    float* tmp = (float*)malloc((size_t)sw * dh * 3 * sizeof(float));
    int maxw = (int)(2.0f * ((float)(sh > sw ? sh : sw)) + 8);
    float* ws = (float*)malloc((size_t)maxw * sizeof(float));
    for (int oy = 0; oy < dh; ++oy) {
        int n, left = sample_weights(oy, sh, dh, ws, &n);
        for (int x = 0; x < sw; ++x)
            for (int c = 0; c < 3; ++c) {
                float t = 0.0f;
                for (int i = 0; i < n; ++i) t += (float)src[((size_t)(left + i) * sw + x) * 3 + c] * ws[i];
                tmp[((size_t)oy * sw + x) * 3 + c] = t;
            }
    }
    for (int ox = 0; ox < dw; ++ox) {
        int n, left = sample_weights(ox, sw, dw, ws, &n);
        for (int y = 0; y < dh; ++y)
            for (int c = 0; c < 3; ++c) {
                float t = 0.0f;
                for (int i = 0; i < n; ++i) t += tmp[((size_t)y * sw + left + i) * 3 + c] * ws[i];
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
                dst[((size_t)y * dw + ox) * 3 + c] = (uint8_t)roundf(t); /* FloatNearest: f32::round */
            }
    }
    free(ws);
    free(tmp);
}

/* yolact.rs:192-217 — unpack, resize_exact(2S,S), crop (0,0,S,S) and (S,0,S,S). */
void orc_classify_pre(const uint32_t* frame, int w, int h, int S, uint8_t* tiles) {
    uint8_t* rgb = (uint8_t*)malloc((size_t)w * h * 3);
    uint8_t* sq = (uint8_t*)malloc((size_t)2 * S * S * 3);
    orc_unpack_rgb(frame, (size_t)w * h, rgb);
    orc_resize_triangle_rgb8(rgb, w, h, sq, 2 * S, S);
    for (int t = 0; t < 2; ++t)
        for (int y = 0; y < S; ++y)
            memcpy(tiles + ((size_t)t * S * S + (size_t)y * S) * 3, sq + ((size_t)y * 2 * S + (size_t)t * S) * 3, (size_t)S * 3);
    free(rgb);
    free(sq);
}

/* yolact.rs:189 + :219-233 — per-tile postprocess, stitch t1|t2 row-wise, unpack the CLASS-CODE
 * image to bytes, resize_exact(w,h,Triangle), repack, overwrite the frame. */
int orc_classify_post(const float* cells2, int S, int nch, int mode, uint32_t* frame, int w, int h) {
    int grid = S / 8;
    uint32_t* t = (uint32_t*)malloc((size_t)2 * S * S * sizeof(uint32_t));
    int div = 0;
    for (int k = 0; k < 2; ++k)
        div |= orc_postprocess_tile(cells2 + (size_t)k * grid * grid * nch, grid, nch, mode, t + (size_t)k * S * S);
    if (!div) {
        uint32_t* st = (uint32_t*)malloc((size_t)2 * S * S * sizeof(uint32_t));
        for (int y = 0; y < S; ++y) {
            memcpy(st + (size_t)y * 2 * S, t + (size_t)y * S, (size_t)S * 4);
            memcpy(st + (size_t)y * 2 * S + S, t + (size_t)S * S + (size_t)y * S, (size_t)S * 4);
        }
        uint8_t* b = (uint8_t*)malloc((size_t)2 * S * S * 3);
        uint8_t* r = (uint8_t*)malloc((size_t)w * h * 3);
        orc_unpack_rgb(st, (size_t)2 * S * S, b);
        orc_resize_triangle_rgb8(b, 2 * S, S, r, w, h);
        orc_pack_rgb(r, (size_t)w * h, frame);
        free(st); free(b); free(r);
    }
    free(t);
    return div;
}

/* scene.rs:93 — ((px << 16) >> 16) as u16 */
void orc_consumer_low16(const uint32_t* frame, size_t n, uint16_t* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (uint16_t)((frame[i] << 16) >> 16);
}
