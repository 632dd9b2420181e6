/*
 * orc_fp8.c — OCP FP8 E4M3 (e4m3fn) encode/decode, CPU restatement. TEST INFRASTRUCTURE.
 *
 * Groundwork for an fp8 convolution path: the format gfx950's MFMA consumes
 * (/opt/skills/guides/cdna_hip_programming.md: "gfx950 uses OCP FP8 (e4m3fn and e5m2)"). Nothing in the
 * reference uses fp8 (its model is uint8-quantised, data/README.md:5-16); BASELINE.json's configs[4]
 * asks for it. PARITY UNPINNED against the reference by construction; the encoder is pinned by an
 * exhaustive nearest-value search over all 65 536 f16 inputs (tests/test_oracle_fp8.py).
 *
 * Format: 1 sign, 4 exponent bits (bias 7), 3 mantissa bits; no infinities; S.1111.111 is NaN; the
 * largest finite value is S.1111.110 = 448; subnormals m/8 * 2^-6.
 * Encode: round to nearest, ties to even, SATURATING at +-448 (an out-of-range product of
 * calibration must not become NaN); NaN in -> 0x7F | sign.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

float orc_e4m3_to_f32(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);               /* m/8 * 2^-6 */
    else v = ldexpf((float)(8 + m), e - 10);            /* (1 + m/8) * 2^(e-7) */
    return s ? -v : v;
}

uint8_t orc_e4m3_from_f32(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if (isnan(x)) return sign | 0x7F;
    float a = fabsf(x);
    if (a >= 448.0f) return sign | 0x7E;                /* saturate (also +-inf) */
    if (a < ldexpf(1.0f, -10)) return sign;             /* below half the smallest subnormal (2^-9): rounds to 0 ... */
    /* quantum of the target binade: subnormal range uses 2^-9, normal numbers 2^(E-3) */
    int ex;
    (void)frexpf(a, &ex);                               /* a = f * 2^ex, f in [0.5, 1) -> exponent E = ex - 1 */
    int E = ex - 1;
    if (E < -6) E = -6;
    const float q = ldexpf(1.0f, E - 3);
    float n = a / q;                                    /* exact: power-of-two division */
    float r = rintf(n);                                 /* default rounding mode: nearest, ties to even */
    (void)n;
    float v = r * q;                                    /* may carry into the next binade, still exact */
    if (v >= 448.0f) return sign | 0x7E;
    if (v == 0.0f) return sign;
    (void)frexpf(v, &ex);
    E = ex - 1;
    if (E < -6) {                                       /* subnormal result */
        const int m = (int)(v / ldexpf(1.0f, -9));
        return (uint8_t)(sign | m);
    }
    const int m = (int)(v / ldexpf(1.0f, E - 3)) - 8;
    return (uint8_t)(sign | ((E + 7) << 3) | m);
}

/* y[i] = encode(x[i] * inv_scale); x already f16-representable when it models an f16 tensor */
void orc_quantize_e4m3(const float* x, long long n, float inv_scale, uint8_t* y) {
    for (long long i = 0; i < n; ++i) y[i] = orc_e4m3_from_f32(x[i] * inv_scale);
}
