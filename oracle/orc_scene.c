/*
 * orc_scene.c — CPU restatement of the reference's scene back-end. TEST INFRASTRUCTURE.
 *
 * Follows /root/reference/shaders/pt_cloud.comp (height map + ball centroids) and
 * /root/reference/shaders/pt_cloud_weights.comp (world positions + 8-neighbour edge lengths), dispatched
 * by /root/reference/src/scene.rs:238-260 as [80,60,1] workgroups of 8x8 over a 640x480 frame.
 *
 * PARITY UNPINNED, and necessarily so: as written the shaders have no single defined result.
 *   - store_ball (pt_cloud.comp:78-82) is a non-atomic read-modify-write shared by every ball pixel;
 *   - barrier() (pt_cloud_weights.comp:89,:113) synchronises one 8x8 workgroup, but stages 2 and 3 read
 *     texels that other workgroups write;
 *   - pow(x, 2) with x < 0 (pt_cloud.comp:68, pt_cloud_weights.comp:49-53) and pow(-1, t) (val = 0,
 *     pt_cloud.comp:62,:69) are undefined in GLSL; uint(NaN) is undefined;
 *   - texture() with normalised coordinates and a Nearest sampler (pt_cloud.comp:88-92) picks texel
 *     floor(x / 640 * 640), which is hardware rounding ("TODO Debug depth_samp input issues", :3);
 *   - the run artefacts map.bmp / depth.bmp are lossy (u32 -> u8 wrap, / 17: src/scene.rs:290-303,:191-194).
 * What is frozen here is the deterministic reading: every stage completes over the whole frame
 * before the next starts; texel (x, y) is read for pixel (x, y); squares are products; a bump whose sigmoid base
 * C_1 = val / 0.1 - 1 is not positive adds nothing; ball centroids are exact integer means.
 * mode 0 (STRICT) keeps the quirks that ARE defined: pack() uses `&` (pt_cloud_weights.comp:32: always 0, so every
 * neighbour "position" is world(0,0)); mode 1 (SANE) measures the distance to the actual neighbour.
 * Every float expression below is one IEEE operation per operator, in this order (-ffp-contract=off), with sqrtf and
 * division correctly rounded, so the HIP kernels (csrc/scene.hip) reproduce it bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define SC_MAX_DEPTH 4000.0f
#define SC_TAN_HALF_YFOV 0.55430907f   /* tan(1.01229096616 / 2), pt_cloud.comp:25 */
#define SC_TAN_HALF_XFOV 0.9489646f   /* tan(1.51843644924 / 2), pt_cloud.comp:26 */
#define SC_BOT_AVOID 100.0f            /* pt_cloud.comp:30 */
#define SC_BOT_NORM 20                 /* :34 */
#define SC_TERRAIN_NORM 10             /* :35 */
#define SC_BUMP_ERR 0.1f               /* :37 */

/* natural logarithm of a positive normal float: range reduction to [sqrt(1/2), sqrt(2)), log(1+f) = 2 atanh(f/(2+f))
 * as an odd series in s = f / (2 + f). Identical source in csrc/yh_internal.h (it is the spec). */
float orc_spec_logf(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    int ex = (int)(b >> 23) - 127;
    b = (b & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &b, 4);
    if (m > 1.41421356f) { m = m * 0.5f; ex += 1; }
    const float f = m - 1.0f, s = f / (2.0f + f), z = s * s;
    float p = fmaf(z, 0.11111111f, 0.14285715f);
    p = fmaf(p, z, 0.2f);
    p = fmaf(p, z, 0.33333334f);
    const float s2 = s + s;
    const float r = fmaf(s2 * z, p, s2);
    return fmaf((float)ex, 0.69314718f, r);
}

/* pow(a, e) for a > 0 as exp(e * log(a)) on the spec functions */
static float spec_powf(float a, float e) { return orc_spec_expf(e * orc_spec_logf(a)); }

/* pt_cloud.comp:44-76 */
static void bump(uint32_t* map, int W, int H, int px, int py, float val, int L) {
    const float C1 = val / SC_BUMP_ERR - 1.0f, C2 = 2.0f / (float)L;
    if (!(C1 > 0.0f)) return;   /* pow(C_1 <= 0, t): undefined in GLSL -> contributes nothing */
    for (int lx = 0; lx < 2 * L; ++lx)
        for (int ly = 0; ly < 2 * L; ++ly) {
            const int x = px - L + lx, y = py - L + ly;
            if (x > 0 && y > 0 && x < W - 1 && y < H - 1) {
                const int dx = px - x, dy = py - y;
                const float prox = sqrtf((float)(dx * dx + dy * dy));
                const float e = C2 * prox - 1.0f;
                const float y_add = val / (1.0f + spec_powf(C1, e));
                const uint32_t v = y_add >= 1.0f ? (uint32_t)y_add : 0u;   /* uint(): truncation; NaN / negative -> 0 */
                if (v > map[(size_t)y * W + x]) map[(size_t)y * W + x] = v;   /* imageAtomicMax */
            }
        }
}

/* depth [H][W] u16; cls_id [H][W][2] u8 (R8G8: class, id); outputs: map [H][W] u32, world / conn0 / conn1 [H][W][4] f32,
 * balls [100][4] f32. mode: 0 STRICT, 1 SANE. */
void orc_scene(const uint16_t* depth, const uint8_t* cls_id, int W, int H, int mode,
               uint32_t* map, float* world, float* conn0, float* conn1, float* balls) {
    memset(map, 0, (size_t)W * H * 4);
    long long bx[100], by[100], bn[100];
    memset(bx, 0, sizeof bx); memset(by, 0, sizeof by); memset(bn, 0, sizeof bn);
    /* ---- pt_cloud.comp main (:84-123) */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            const float ty = SC_TAN_HALF_YFOV * (float)y * 2.0f / (float)H, tx = SC_TAN_HALF_XFOV * (float)x * 2.0f / (float)W;
            const float cy = 1.0f / sqrtf(1.0f + ty * ty), cx = 1.0f / sqrtf(1.0f + tx * tx);   /* cos(atan(t)) */
            const float d = (float)depth[i] * cy * cx;
            const int dic = (int)((float)H * d / SC_MAX_DEPTH);
            const int cls = cls_id[2 * i], id = cls_id[2 * i + 1];
            int action = cls;
            if (action > 1) action = action - 1;
            const int nx = x, ny = H - dic;
            if (action == 0) bump(map, W, H, nx, ny, (float)y, SC_TERRAIN_NORM);
            else if (action == 2) { if (id < 100) { bx[id] += nx; by[id] += ny; bn[id] += 1; } }
            else bump(map, W, H, nx, ny, SC_BOT_AVOID, SC_BOT_NORM);
        }
    for (int k = 0; k < 100; ++k) {
        balls[4 * k + 0] = bn[k] ? (float)((double)bx[k] / (double)bn[k]) : 0.0f;
        balls[4 * k + 1] = bn[k] ? (float)((double)by[k] / (double)bn[k]) : 0.0f;
        balls[4 * k + 2] = (float)bn[k];
        balls[4 * k + 3] = 0.0f;
    }
    /* ---- pt_cloud_weights.comp, stage 1 (:57-87): world position, encoded position */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            world[4 * i + 0] = (float)x; world[4 * i + 1] = (float)map[i]; world[4 * i + 2] = (float)y; world[4 * i + 3] = 0.0f;
        }
    /* stage 2 (:91-111): distance to the position each of four neighbours published */
    static const int n2[4][2] = { { 0, 1 }, { -1, 1 }, { -1, 0 }, { -1, -1 } };   /* r, g, b, a */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            for (int k = 0; k < 4; ++k) {
                const int qx = x + n2[k][0], qy = y + n2[k][1];
                float v = -1.0f;
                if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                    /* STRICT: pack(x, y) = float((x << 16) & y) = 0 for every pixel -> unpack -> world(0, 0) */
                    const size_t j = mode == 0 ? 0 : (size_t)qy * W + qx;
                    const float ddx = world[4 * i] - world[4 * j], ddy = world[4 * i + 1] - world[4 * j + 1], ddz = world[4 * i + 2] - world[4 * j + 2];
                    v = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                }
                conn1[4 * i + k] = v;
            }
        }
    /* stage 3 (:115-123): the other four neighbours' view of this pixel */
    static const int n3[4][2] = { { 0, -1 }, { 1, -1 }, { 1, 0 }, { 1, 1 } };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            for (int k = 0; k < 4; ++k) {
                const int qx = x + n3[k][0], qy = y + n3[k][1];
                conn0[4 * i + k] = (qx >= 0 && qx < W && qy >= 0 && qy < H) ? conn1[4 * ((size_t)qy * W + qx) + k] : -1.0f;
            }
        }
}
