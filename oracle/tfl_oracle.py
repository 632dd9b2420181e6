"""numpy restatement of the TensorFlow Lite reference kernels for the op set of the reference's
model (data/FRC_model_edgetpu.log:7-19) — TEST INFRASTRUCTURE ONLY (tests/, never the product).

PARITY UNPINNED: the arithmetic lives in TensorFlow Lite's C++ interpreter, reached through the
un-vendored crate tflite 0.9.0 (git littletitan/tflite-rs@abcaeab4, /root/reference/Cargo.lock:
1106-1108; call sites src/yolact.rs:18-35, :149-181), and the model file itself is absent
(.MISSING_LARGE_BLOBS:1-2). What follows restates TFLite's PUBLISHED uint8 (asymmetric, per-tensor)
reference semantics: gemmlowp fixed-point requantisation (QuantizeMultiplier,
SaturatingRoundingDoublingHighMul, RoundingDivideByPOT), quantised CONV_2D / DEPTHWISE_CONV_2D /
ADD / PAD / RESIZE_BILINEAR / TANH (256-entry table) / RELU / CONCATENATION / QUANTIZE /
DEQUANTIZE / RESHAPE. It evaluates a model given as Python objects (tests/tfl_builder.Model), i.e.
it never goes through the product's flatbuffer reader: agreement between the two validates both.
"""
import math

import numpy as np

I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


def _f32(s):
    """A tensor's scale is a float32 in the flatbuffer; TFLite promotes THAT value to double
    (kernel_util.cc: static_cast<double>(input->params.scale)), so every multiplier below starts
    from the float32-rounded scale, whatever precision the caller's Python float carries."""
    return float(np.float32(s))


def quantize_multiplier(m):
    """tflite::QuantizeMultiplier(double): (int32 Q31 mantissa, shift)."""
    if m == 0.0:
        return 0, 0
    q, shift = math.frexp(m)
    qf = int(math.floor(q * (1 << 31) + 0.5))  # TfLiteRound
    if qf == (1 << 31):
        qf //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    return qf, shift


def _srdhm(a, b):
    """gemmlowp SaturatingRoundingDoublingHighMul on int32 arrays (b scalar)."""
    a = np.asarray(a, np.int64)
    ab = a * np.int64(b)
    nudge = np.where(ab >= 0, np.int64(1 << 30), np.int64(1 - (1 << 30)))
    q = ab + nudge
    r = np.where(q >= 0, q >> 31, -((-q) >> 31))          # C++ '/' truncates toward zero
    sat = (a == I32_MIN) & (b == I32_MIN)
    return np.where(sat, I32_MAX, r).astype(np.int64)


def _rdbpot(x, e):
    """gemmlowp RoundingDivideByPOT."""
    x = np.asarray(x, np.int64)
    if e == 0:
        return x
    mask = np.int64((1 << e) - 1)
    rem = x & mask
    thr = (mask >> 1) + (x < 0)
    return (x >> e) + (rem > thr)


def mbqm(x, mult, shift):
    """tflite::MultiplyByQuantizedMultiplier."""
    left, right = (shift, 0) if shift > 0 else (0, -shift)
    return _rdbpot(_srdhm(np.asarray(x, np.int64) * (1 << left), mult), right)


def act_range_u8(act, scale, zp):
    """CalculateActivationRangeQuantized for uint8: act 0 none, 1 relu, 3 relu6."""
    def q(v):
        d = float(np.float32(v) / np.float32(scale))          # f32 division, TfLiteRound
        return zp + int(math.floor(d + 0.5) if d >= 0 else math.ceil(d - 0.5))
    lo, hi = 0, 255
    if act == 1:
        lo = max(lo, q(0.0))
    elif act == 3:
        lo, hi = max(lo, q(0.0)), min(hi, q(6.0))
    elif act == 2:
        lo, hi = max(lo, q(-1.0)), min(hi, q(1.0))
    return lo, hi


def _same_pad(inp, k, stride, dil):
    out = (inp + stride - 1) // stride
    eff = (k - 1) * dil + 1
    total = max((out - 1) * stride + eff - inp, 0)
    return out, total // 2


def _geom(h, w, kh, kw, sh, sw, dh, dw, padding):
    if padding == 0:  # SAME
        ho, ph = _same_pad(h, kh, sh, dh)
        wo, pw = _same_pad(w, kw, sw, dw)
    else:             # VALID
        ho = (h - ((kh - 1) * dh + 1) + sh) // sh
        wo = (w - ((kw - 1) * dw + 1) + sw) // sw
        ph = pw = 0
    return ho, wo, ph, pw


def conv2d_u8(x, zx, sx, w, zw, sw_, bias, zo, so, stride, padding, act, dil=(1, 1)):
    """reference_ops::Conv (uint8). x [1,H,W,Ci], w [Co,kh,kw,Ci] (OHWI), bias int32 [Co]."""
    _, h, ww, ci = x.shape
    co, kh, kw, _ = w.shape
    ho, wo, ph, pw = _geom(h, ww, kh, kw, stride[0], stride[1], dil[0], dil[1], padding)
    mult, shift = quantize_multiplier(_f32(sx) * _f32(sw_) / _f32(so))
    lo, hi = act_range_u8(act, so, zo)
    xi = x[0].astype(np.int64) - zx
    wi = w.astype(np.int64) - zw
    acc = np.zeros((ho, wo, co), np.int64)
    for r in range(kh):
        for s in range(kw):
            for oy in range(ho):
                iy = oy * stride[0] - ph + r * dil[0]
                if iy < 0 or iy >= h:
                    continue
                ix = np.arange(wo) * stride[1] - pw + s * dil[1]
                ok = (ix >= 0) & (ix < ww)
                if not ok.any():
                    continue
                acc[oy, ok, :] += xi[iy, ix[ok], :] @ wi[:, r, s, :].T
    acc += bias.astype(np.int64)[None, None, :]
    out = mbqm(acc, mult, shift) + zo
    return np.clip(out, lo, hi).astype(np.uint8)[None]


def dwconv2d_u8(x, zx, sx, w, zw, sw_, bias, zo, so, stride, padding, act, depth_multiplier=1, dil=(1, 1)):
    """reference_ops::DepthwiseConv (uint8). w [1,kh,kw,Ci*dm]; out channel = ic*dm + m."""
    _, h, ww, ci = x.shape
    _, kh, kw, co = w.shape
    ho, wo, ph, pw = _geom(h, ww, kh, kw, stride[0], stride[1], dil[0], dil[1], padding)
    mult, shift = quantize_multiplier(_f32(sx) * _f32(sw_) / _f32(so))
    lo, hi = act_range_u8(act, so, zo)
    xi = x[0].astype(np.int64) - zx
    wi = w[0].astype(np.int64) - zw
    src = np.repeat(np.arange(ci), depth_multiplier)
    acc = np.zeros((ho, wo, co), np.int64)
    for r in range(kh):
        for s in range(kw):
            for oy in range(ho):
                iy = oy * stride[0] - ph + r * dil[0]
                if iy < 0 or iy >= h:
                    continue
                ix = np.arange(wo) * stride[1] - pw + s * dil[1]
                ok = (ix >= 0) & (ix < ww)
                acc[oy, ok, :] += xi[iy, ix[ok], :][:, src] * wi[r, s, :][None, :]
    acc += bias.astype(np.int64)[None, None, :]
    out = mbqm(acc, mult, shift) + zo
    return np.clip(out, lo, hi).astype(np.uint8)[None]


def add_u8(a, za, sa, b, zb, sb, zo, so, act):
    """reference_ops::Add (uint8), left_shift = 20."""
    ls = 20
    twice = 2.0 * max(_f32(sa), _f32(sb))
    m1, s1 = quantize_multiplier(_f32(sa) / twice)
    m2, s2 = quantize_multiplier(_f32(sb) / twice)
    mo, so_ = quantize_multiplier(twice / ((1 << ls) * _f32(so)))
    lo, hi = act_range_u8(act, so, zo)
    v1 = mbqm((a.astype(np.int64) - za) * (1 << ls), m1, s1)
    v2 = mbqm((b.astype(np.int64) - zb) * (1 << ls), m2, s2)
    out = mbqm(v1 + v2, mo, so_) + zo
    return np.clip(out, lo, hi).astype(np.uint8)


def requant_u8(x, zi, si, zo, so):
    """QUANTIZE uint8 -> uint8 (reference_ops::Requantize)."""
    m, s = quantize_multiplier(_f32(si) / _f32(so))
    return np.clip(mbqm(x.astype(np.int64) - zi, m, s) + zo, 0, 255).astype(np.uint8)


def quantize_f32(x, zo, so):
    """QUANTIZE float -> uint8 (AffineQuantize): round(x / scale) + zp, f32 division, TfLiteRound."""
    v = (x.astype(np.float32) / np.float32(so)).astype(np.float32)
    r = np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5)))
    return np.clip(r.astype(np.int64) + zo, 0, 255).astype(np.uint8)


def dequantize_u8(x, z, s):
    return (np.float32(s) * (x.astype(np.int32) - z).astype(np.float32)).astype(np.float32)


def relu_u8(x, zi, si, zo, so, act=1):
    m, s = quantize_multiplier(_f32(si) / _f32(so))
    lo, hi = act_range_u8(act, so, zo)
    return np.clip(mbqm(x.astype(np.int64) - zi, m, s) + zo, lo, hi).astype(np.uint8)


def tanh_lut_u8(zi, si, zo, so):
    """256-entry table (PopulateLookupTable<uint8_t>): f32 dequantise, tanh, round(y * (1/scale)) + zp."""
    inv = np.float32(1.0) / np.float32(so)
    lut = np.zeros(256, np.uint8)
    for q in range(256):
        x = np.float32(si) * np.float32(q - zi)
        y = np.float32(math.tanh(float(x)))
        r = float(np.float32(y * inv))
        rr = math.floor(r + 0.5) if r >= 0 else math.ceil(r - 0.5)
        lut[q] = min(255, max(0, int(rr) + zo))
    return lut


def pad_u8(x, pads, zo):
    return np.pad(x, [(int(a), int(b)) for a, b in pads], constant_values=np.uint8(zo))


def resize_bilinear_u8(x, ho, wo, align_corners=False, half_pixel=False):
    """optimized_ops-style uint8 resize: f32 interpolation, result = (uint8)(v + 0.5f)."""
    _, h, w, c = x.shape
    def scale(i, o):
        return np.float32((i - 1) / (o - 1)) if (align_corners and o > 1) else np.float32(i) / np.float32(o)
    hs, ws = scale(h, ho), scale(w, wo)
    out = np.zeros((1, ho, wo, c), np.uint8)
    xf = x[0].astype(np.float32)
    for oy in range(ho):
        iy = (np.float32(oy) + np.float32(0.5)) * hs - np.float32(0.5) if half_pixel else np.float32(oy) * hs
        y0 = max(int(math.floor(float(iy))), 0)
        y1 = min(int(math.ceil(float(iy))), h - 1)
        fy = np.float32(iy - np.float32(y0))
        for ox in range(wo):
            ix = (np.float32(ox) + np.float32(0.5)) * ws - np.float32(0.5) if half_pixel else np.float32(ox) * ws
            x0 = max(int(math.floor(float(ix))), 0)
            x1 = min(int(math.ceil(float(ix))), w - 1)
            fx = np.float32(ix - np.float32(x0))
            one = np.float32(1.0)
            v = (xf[y0, x0] * (one - fy) * (one - fx) + xf[y1, x0] * fy * (one - fx) +
                 xf[y0, x1] * (one - fy) * fx + xf[y1, x1] * fy * fx).astype(np.float32)
            out[0, oy, ox] = np.clip(np.floor(v + np.float32(0.5)), 0, 255).astype(np.uint8)
    return out


def concat_u8(xs, params, zo, so, axis):
    """ConcatenationWithScaling: inputs with other (scale, zp) are rescaled in f32 with round()."""
    outs = []
    inv = np.float32(1.0) / np.float32(so)
    for x, (z, s) in zip(xs, params):
        if z == zo and np.float32(s) == np.float32(so):
            outs.append(x)
        else:
            sc = np.float32(s) * inv
            bias = np.float32(-z) * sc
            v = (x.astype(np.float32) * sc + bias).astype(np.float32)
            r = np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64) + zo
            outs.append(np.clip(r, 0, 255).astype(np.uint8))
    return np.concatenate(outs, axis=axis)


def run_model(model, inputs):
    """Evaluates tests/tfl_builder.Model on a dict {tensor index: array}; returns all tensor values."""
    val = {}
    for i, t in enumerate(model.tensors):
        if t.data is not None:
            val[i] = t.data
    val.update(inputs)
    T = model.tensors
    for op in model.ops:
        ins, outs, o = op.inputs, op.outputs, op.opts
        t_out = T[outs[0]]
        if op.code == "CONV_2D":
            x, w, b = T[ins[0]], T[ins[1]], T[ins[2]]
            val[outs[0]] = conv2d_u8(val[ins[0]], x.zp, x.scale, val[ins[1]], w.zp, w.scale, val[ins[2]], t_out.zp, t_out.scale,
                                     (o["stride_h"], o["stride_w"]), o["padding"], o.get("act", 0))
        elif op.code == "DEPTHWISE_CONV_2D":
            x, w = T[ins[0]], T[ins[1]]
            val[outs[0]] = dwconv2d_u8(val[ins[0]], x.zp, x.scale, val[ins[1]], w.zp, w.scale, val[ins[2]], t_out.zp, t_out.scale,
                                       (o["stride_h"], o["stride_w"]), o["padding"], o.get("act", 0), o.get("depth_multiplier", 1))
        elif op.code == "ADD":
            a, b = T[ins[0]], T[ins[1]]
            val[outs[0]] = add_u8(val[ins[0]], a.zp, a.scale, val[ins[1]], b.zp, b.scale, t_out.zp, t_out.scale, o.get("act", 0))
        elif op.code == "PAD":
            val[outs[0]] = pad_u8(val[ins[0]], val[ins[1]], t_out.zp)
        elif op.code == "RESIZE_BILINEAR":
            size = val[ins[1]]
            val[outs[0]] = resize_bilinear_u8(val[ins[0]], int(size[0]), int(size[1]), o.get("align_corners", False), o.get("half_pixel_centers", False))
        elif op.code == "TANH":
            x = T[ins[0]]
            val[outs[0]] = tanh_lut_u8(x.zp, x.scale, t_out.zp, t_out.scale)[val[ins[0]]]
        elif op.code == "RELU":
            x = T[ins[0]]
            val[outs[0]] = relu_u8(val[ins[0]], x.zp, x.scale, t_out.zp, t_out.scale)
        elif op.code == "CONCATENATION":
            axis = o["axis"] if o["axis"] >= 0 else o["axis"] + len(t_out.shape)
            val[outs[0]] = concat_u8([val[i] for i in ins], [(T[i].zp, T[i].scale) for i in ins], t_out.zp, t_out.scale, axis)
        elif op.code == "RESHAPE":
            val[outs[0]] = val[ins[0]].reshape(t_out.shape)
        elif op.code == "QUANTIZE":
            x = T[ins[0]]
            val[outs[0]] = quantize_f32(val[ins[0]], t_out.zp, t_out.scale) if x.dtype == "f32" else \
                requant_u8(val[ins[0]], x.zp, x.scale, t_out.zp, t_out.scale)
        elif op.code == "DEQUANTIZE":
            x = T[ins[0]]
            val[outs[0]] = dequantize_u8(val[ins[0]], x.zp, x.scale)
        else:
            raise NotImplementedError(op.code)
    return val
