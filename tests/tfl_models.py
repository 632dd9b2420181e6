"""Synthetic TFLite models for tests (tests/tfl_builder.py objects)."""
import numpy as np

from tfl_builder import Model, Op, T


def _q(rng, shape, lo=0, hi=256):
    return rng.integers(lo, hi, shape, dtype=np.uint8)


def single_op(code, rng, **kw):
    m = Model()
    if code in ("CONV_2D", "DEPTHWISE_CONV_2D"):
        h, w, ci, co, k = kw.get("h", 9), kw.get("w", 7), kw.get("ci", 8), kw.get("co", 12), kw.get("k", 3)
        stride, padding, act, dm = kw.get("stride", 1), kw.get("padding", 0), kw.get("act", 0), kw.get("dm", 1)
        x = m.add(T("x", (1, h, w, ci), "u8", scale=0.05, zp=int(rng.integers(100, 150))))
        if code == "CONV_2D":
            wt = m.add(T("w", (co, k, k, ci), "u8", data=_q(rng, (co, k, k, ci)), scale=0.01, zp=int(rng.integers(100, 160))))
        else:
            co = ci * dm
            wt = m.add(T("w", (1, k, k, co), "u8", data=_q(rng, (1, k, k, co)), scale=0.02, zp=int(rng.integers(100, 160))))
        b = m.add(T("b", (co,), "i32", data=rng.integers(-2000, 2000, co), scale=0.0005, zp=0))
        if padding == 0:
            ho, wo = -(-h // stride), -(-w // stride)
        else:
            ho, wo = (h - k + stride) // stride, (w - k + stride) // stride
        y = m.add(T("y", (1, ho, wo, co), "u8", scale=kw.get("so", 0.08), zp=int(rng.integers(0, 140))))
        opts = dict(padding=padding, stride_w=stride, stride_h=stride, act=act)
        if code == "DEPTHWISE_CONV_2D":
            opts["depth_multiplier"] = dm
        m.ops.append(Op(code, [x, wt, b], [y], **opts))
    elif code == "ADD":
        shp = (1, 6, 5, 16)
        x = m.add(T("a", shp, "u8", scale=0.04, zp=120))
        c = m.add(T("c", shp, "u8", data=_q(rng, shp), scale=0.07, zp=100))
        y = m.add(T("y", shp, "u8", scale=0.09, zp=110))
        m.ops.append(Op("ADD", [x, c], [y], act=kw.get("act", 0)))
    elif code == "PAD":
        x = m.add(T("x", (1, 5, 6, 4), "u8", scale=0.1, zp=77))
        pd = m.add(T("p", (4, 2), "i32", data=[[0, 0], [1, 2], [2, 1], [0, 0]]))
        y = m.add(T("y", (1, 8, 9, 4), "u8", scale=0.1, zp=77))
        m.ops.append(Op("PAD", [x, pd], [y]))
    elif code == "RESIZE_BILINEAR":
        hi, wi, ho, wo = kw.get("hi", 5), kw.get("wi", 4), kw.get("ho", 10), kw.get("wo", 8)
        x = m.add(T("x", (1, hi, wi, 6), "u8", scale=0.1, zp=3))
        sz = m.add(T("s", (2,), "i32", data=[ho, wo]))
        y = m.add(T("y", (1, ho, wo, 6), "u8", scale=0.1, zp=3))
        m.ops.append(Op("RESIZE_BILINEAR", [x, sz], [y], align_corners=kw.get("align_corners", False), half_pixel_centers=kw.get("half_pixel_centers", False)))
    elif code == "TANH":
        x = m.add(T("x", (1, 4, 4, 16), "u8", scale=0.03, zp=128))
        y = m.add(T("y", (1, 4, 4, 16), "u8", scale=1.0 / 128, zp=128))
        m.ops.append(Op("TANH", [x], [y]))
    elif code == "RELU":
        x = m.add(T("x", (1, 4, 4, 16), "u8", scale=0.05, zp=120))
        y = m.add(T("y", (1, 4, 4, 16), "u8", scale=0.03, zp=10))
        m.ops.append(Op("RELU", [x], [y]))
    elif code == "QUANTIZE":
        x = m.add(T("x", (1, 4, 4, 16), "u8", scale=0.05, zp=120))
        y = m.add(T("y", (1, 4, 4, 16), "u8", scale=0.11, zp=30))
        m.ops.append(Op("QUANTIZE", [x], [y]))
    elif code == "CONCATENATION":
        x = m.add(T("x", (1, 3, 3, 4), "u8", scale=0.05, zp=120))
        c = m.add(T("c", (1, 3, 3, 6), "u8", data=_q(rng, (1, 3, 3, 6)), scale=0.08, zp=90))
        d = m.add(T("d", (1, 3, 3, 2), "u8", data=_q(rng, (1, 3, 3, 2)), scale=0.05, zp=120))
        y = m.add(T("y", (1, 3, 3, 12), "u8", scale=0.05, zp=120))
        m.ops.append(Op("CONCATENATION", [x, c, d], [y], axis=kw.get("axis", 3)))
    elif code == "RESHAPE":
        x = m.add(T("x", (1, 4, 4, 6), "u8", scale=0.05, zp=12))
        y = m.add(T("y", (1, 16, 6), "u8", scale=0.05, zp=12))
        m.ops.append(Op("RESHAPE", [x], [y], new_shape=[1, 16, 6]))
    elif code == "DEQUANTIZE":
        x = m.add(T("x", (1, 4, 4, 6), "u8", scale=0.0625, zp=128))
        y = m.add(T("y", (1, 4, 4, 6), "f32"))
        m.ops.append(Op("DEQUANTIZE", [x], [y]))
    m.inputs, m.outputs = [0], [len(m.tensors) - 1]
    return m


def mobilenet_like(rng, S=64, C=6):
    """A small MobileNetV2-FPN-YOLACT-shaped graph with the reference model's op mix: stem conv s2,
    inverted-residual blocks (expand 1x1 / depthwise 3x3 / project 1x1 / ADD), PAD + VALID conv,
    RESIZE_BILINEAR top-down + ADD, RELU, TANH, QUANTIZE, CONCATENATION, RESHAPE; five outputs, output
    4 = [1, (S/8)^2, C] class logits (what src/yolact.rs:91 reads)."""
    m = Model()
    sc = lambda: float(rng.uniform(0.02, 0.08))
    zp = lambda: int(rng.integers(90, 160))

    def conv(x, ci, co, k, stride, hw, act=1, padding=0, so=None):
        w = m.add(T(f"w{len(m.tensors)}", (co, k, k, ci), "u8", data=_q(rng, (co, k, k, ci), 96, 160), scale=0.004, zp=128))
        b = m.add(T(f"b{len(m.tensors)}", (co,), "i32", data=rng.integers(-300, 300, co), scale=0.0002, zp=0))
        ho = -(-hw // stride) if padding == 0 else (hw - k + stride) // stride
        y = m.add(T(f"c{len(m.tensors)}", (1, ho, ho, co), "u8", scale=so or sc(), zp=zp()))
        m.ops.append(Op("CONV_2D", [x, w, b], [y], padding=padding, stride_w=stride, stride_h=stride, act=act))
        return y, ho

    def dw(x, c, stride, hw):
        w = m.add(T(f"dw{len(m.tensors)}", (1, 3, 3, c), "u8", data=_q(rng, (1, 3, 3, c), 64, 192), scale=0.01, zp=128))
        b = m.add(T(f"db{len(m.tensors)}", (c,), "i32", data=rng.integers(-300, 300, c), scale=0.0005, zp=0))
        ho = -(-hw // stride)
        y = m.add(T(f"d{len(m.tensors)}", (1, ho, ho, c), "u8", scale=sc(), zp=zp()))
        m.ops.append(Op("DEPTHWISE_CONV_2D", [x, w, b], [y], padding=0, stride_w=stride, stride_h=stride, act=3, depth_multiplier=1))
        return y, ho

    def block(x, c, hw):
        e, _ = conv(x, c, c * 2, 1, 1, hw, act=3)
        d, _ = dw(e, c * 2, 1, hw)
        p, _ = conv(d, c * 2, c, 1, 1, hw, act=0)
        y = m.add(T(f"a{len(m.tensors)}", (1, hw, hw, c), "u8", scale=sc(), zp=zp()))
        m.ops.append(Op("ADD", [x, p], [y], act=0))
        return y

    x = m.add(T("input", (1, S, S, 3), "u8", scale=1 / 128, zp=128))
    s1, hw = conv(x, 3, 8, 3, 2, S)              # S/2
    d1, hw = dw(s1, 8, 2, hw)                    # S/4
    c2, _ = conv(d1, 8, 16, 1, 1, hw, act=0)
    c2 = block(c2, 16, hw)
    d2, hw8 = dw(c2, 16, 2, hw)                  # S/8
    c3, _ = conv(d2, 16, 24, 1, 1, hw8, act=0)
    c3 = block(c3, 24, hw8)
    # PAD + VALID stride-2 conv (the converter's explicit-padding pattern) -> S/16
    pd = m.add(T("pads", (4, 2), "i32", data=[[0, 0], [0, 1], [0, 1], [0, 0]]))
    c3t = m.tensors[c3]
    padded = m.add(T("padded", (1, hw8 + 1, hw8 + 1, 24), "u8", scale=c3t.scale, zp=c3t.zp))
    m.ops.append(Op("PAD", [c3, pd], [padded]))
    c4, hw16 = conv(padded, 24, 24, 3, 2, hw8 + 1, act=1, padding=1)
    assert hw16 == hw8 // 2
    # FPN top-down: resize c4 to S/8, add to a lateral of c3, RELU
    sz = m.add(T("size", (2,), "i32", data=[hw8, hw8]))
    c4t = m.tensors[c4]
    up = m.add(T("up", (1, hw8, hw8, 24), "u8", scale=c4t.scale, zp=c4t.zp))
    m.ops.append(Op("RESIZE_BILINEAR", [c4, sz], [up], align_corners=False, half_pixel_centers=False))
    lat, _ = conv(c3, 24, 24, 1, 1, hw8, act=0)
    p3 = m.add(T("p3sum", (1, hw8, hw8, 24), "u8", scale=sc(), zp=zp()))
    m.ops.append(Op("ADD", [lat, up], [p3], act=0))
    p3r = m.add(T("p3", (1, hw8, hw8, 24), "u8", scale=0.04, zp=0))
    m.ops.append(Op("RELU", [p3], [p3r]))
    # heads on p3
    loc, _ = conv(p3r, 24, 4, 3, 1, hw8, act=0)
    conf, _ = conv(p3r, 24, C, 3, 1, hw8, act=0, so=0.006)
    maskc, _ = conv(p3r, 24, 8, 3, 1, hw8, act=0)
    mt = m.add(T("mask_tanh", (1, hw8, hw8, 8), "u8", scale=1 / 128, zp=128))
    m.ops.append(Op("TANH", [maskc], [mt]))
    proto, _ = conv(p3r, 24, 8, 1, 1, hw8, act=1)
    conf_q = m.add(T("conf_q", (1, hw8, hw8, C), "u8", scale=0.0078125, zp=128))
    m.ops.append(Op("QUANTIZE", [conf], [conf_q]))
    cells = m.add(T("cells", (1, hw8 * hw8, C), "u8", scale=0.0078125, zp=128))
    m.ops.append(Op("RESHAPE", [conf_q], [cells], new_shape=[1, hw8 * hw8, C]))
    both = m.add(T("loc_mask", (1, hw8, hw8, 12), "u8", scale=1 / 128, zp=128))
    m.ops.append(Op("CONCATENATION", [loc, mt], [both], axis=3))
    m.inputs = [x]
    m.outputs = [loc, both, mt, proto, cells]
    return m
