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


def mobilenetv2_yolact(rng, S=224, C=81, fpn=128, nproto=32):
    """A full-size stand-in for the reference's data/FRC_model.tflite (absent; op census in
    data/FRC_model_edgetpu.log:5-19, backbone per data/README.md:5-16): MobileNetV2 (width 1.0, 15
    depthwise convs, 9 residual ADDs) + 3-level FPN with two extra stride-2 levels (PAD + VALID convs)
    + protonet (one RESIZE_BILINEAR, one RELU) + a prediction head per level (TANH on the mask
    coefficients), QUANTIZE / RESHAPE / CONCATENATION over the levels; five outputs, output 4 =
    [1, (S/8)^2, C] class logits of the S/8 level (what src/yolact.rs:91 reads). Weights are random:
    the graph is for executor parity against the numpy oracle and for timing, not for accuracy."""
    m = Model()
    sc = lambda: float(rng.uniform(0.02, 0.06))
    zp = lambda: int(rng.integers(100, 150))

    def conv(x, ci, co, k, stride, hw, act=3, padding=0, so=None, zo=None):
        ws = 0.5 / (128.0 * np.sqrt(k * k * ci))          # keeps the accumulators in a sensible range
        w = m.add(T(f"w{len(m.tensors)}", (co, k, k, ci), "u8", data=_q(rng, (co, k, k, ci), 64, 192), scale=float(ws * 8), zp=128))
        b = m.add(T(f"b{len(m.tensors)}", (co,), "i32", data=rng.integers(-200, 200, co), scale=0.0002, zp=0))
        ho = -(-hw // stride) if padding == 0 else (hw - k + stride) // stride
        y = m.add(T(f"c{len(m.tensors)}", (1, ho, ho, co), "u8", scale=so or sc(), zp=zp() if zo is None else zo))
        m.ops.append(Op("CONV_2D", [x, w, b], [y], padding=padding, stride_w=stride, stride_h=stride, act=act))
        return y, ho

    def dw(x, c, stride, hw):
        w = m.add(T(f"dw{len(m.tensors)}", (1, 3, 3, c), "u8", data=_q(rng, (1, 3, 3, c), 64, 192), scale=0.01, zp=128))
        b = m.add(T(f"db{len(m.tensors)}", (c,), "i32", data=rng.integers(-300, 300, c), scale=0.0005, zp=0))
        ho = -(-hw // stride)
        y = m.add(T(f"d{len(m.tensors)}", (1, ho, ho, c), "u8", scale=sc(), zp=zp()))
        m.ops.append(Op("DEPTHWISE_CONV_2D", [x, w, b], [y], padding=0, stride_w=stride, stride_h=stride, act=3, depth_multiplier=1))
        return y, ho

    def add(a, b, shape):
        y = m.add(T(f"a{len(m.tensors)}", shape, "u8", scale=sc(), zp=zp()))
        m.ops.append(Op("ADD", [a, b], [y], act=0))
        return y

    def pad_valid_s2(x, c, co, hw):
        """explicit PAD (0,1) + VALID 3x3 stride-2 conv: the converter's pattern for stride-2 SAME."""
        xt = m.tensors[x]
        pd = m.add(T(f"pads{len(m.tensors)}", (4, 2), "i32", data=[[0, 0], [0, 1], [0, 1], [0, 0]]))
        p = m.add(T(f"padded{len(m.tensors)}", (1, hw + 1, hw + 1, c), "u8", scale=xt.scale, zp=xt.zp))
        m.ops.append(Op("PAD", [x, pd], [p]))
        return conv(p, c, co, 3, 2, hw + 1, act=1, padding=1)

    x = m.add(T("input", (1, S, S, 3), "u8", scale=1 / 128, zp=128))
    y, hw = conv(x, 3, 32, 3, 2, S)                                   # S/2
    y, hw = dw(y, 32, 1, hw)
    y, _ = conv(y, 32, 16, 1, 1, hw, act=0)
    cin, feats = 16, {}
    for t, c, n, s in ((6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 2, 2)):
        for i in range(n):
            stride = s if i == 0 else 1
            e, _ = conv(y, cin, cin * t, 1, 1, hw)
            d, hw2 = dw(e, cin * t, stride, hw)
            p, _ = conv(d, cin * t, c, 1, 1, hw2, act=0)
            y = add(y, p, (1, hw2, hw2, c)) if (stride == 1 and cin == c) else p
            cin, hw = c, hw2
        feats[hw] = (y, cin)                                          # last feature at each resolution
    h8, h16, h32 = S // 8, S // 16, S // 32
    # FPN: laterals, top-down (2 x RESIZE_BILINEAR + ADD), 3x3 output convs, two extra stride-2 levels
    lat = {h: conv(feats[h][0], feats[h][1], fpn, 1, 1, h, act=0)[0] for h in (h8, h16, h32)}

    def up(src, hs, hd):
        st = m.tensors[src]
        sz = m.add(T(f"size{len(m.tensors)}", (2,), "i32", data=[hd, hd]))
        u = m.add(T(f"up{len(m.tensors)}", (1, hd, hd, st.shape[3]), "u8", scale=st.scale, zp=st.zp))
        m.ops.append(Op("RESIZE_BILINEAR", [src, sz], [u], align_corners=False, half_pixel_centers=False))
        return u
    t16 = add(lat[h16], up(lat[h32], h32, h16), (1, h16, h16, fpn))
    t8 = add(lat[h8], up(t16, h16, h8), (1, h8, h8, fpn))
    levels = [(conv(t8, fpn, fpn, 3, 1, h8, act=1)[0], h8), (conv(t16, fpn, fpn, 3, 1, h16, act=1)[0], h16),
              (conv(lat[h32], fpn, fpn, 3, 1, h32, act=1)[0], h32)]
    p6, h64 = pad_valid_s2(levels[2][0], fpn, fpn, h32)
    p7, h128 = pad_valid_s2(p6, fpn, fpn, h64)
    levels += [(p6, h64), (p7, h128)]
    # protonet on the S/8 level: 3 x conv3x3, x2 upsample, conv3x3, 1x1 -> prototypes, RELU
    pr = levels[0][0]
    for _ in range(3):
        pr, _ = conv(pr, fpn, fpn, 3, 1, h8, act=1)
    pr = up(pr, h8, 2 * h8)
    pr, _ = conv(pr, fpn, fpn, 3, 1, 2 * h8, act=1)
    pr, _ = conv(pr, fpn, nproto, 1, 1, 2 * h8, act=0)
    proto = m.add(T("proto", (1, 2 * h8, 2 * h8, nproto), "u8", scale=0.05, zp=0))
    m.ops.append(Op("RELU", [pr], [proto]))
    # prediction head per level (weights unrolled per level, as a converter does with a shared head)
    locs, confs, masks = [], [], []
    for li, (f, h) in enumerate(levels):
        t, _ = conv(f, fpn, fpn, 3, 1, h, act=1)
        for name, co, dst, q_scale in (("loc", 12, locs, 1 / 64), ("conf", 3 * C, confs, 0.0625), ("mask", 3 * nproto, masks, 1 / 128)):
            o, _ = conv(t, fpn, co, 3, 1, h, act=0)
            if name == "mask":
                th = m.add(T(f"tanh{li}", (1, h, h, co), "u8", scale=1 / 128, zp=128))
                m.ops.append(Op("TANH", [o], [th]))
                o = th
            q = m.add(T(f"{name}_q{li}", (1, h, h, co), "u8", scale=q_scale, zp=128))
            m.ops.append(Op("QUANTIZE", [o], [q]))
            r = m.add(T(f"{name}_r{li}", (1, h * h * 3, co // 3), "u8", scale=q_scale, zp=128))
            m.ops.append(Op("RESHAPE", [q], [r], new_shape=[1, h * h * 3, co // 3]))
            dst.append(r)
    npri = sum(h * h * 3 for _, h in levels)
    outs = []
    for name, parts, last in (("loc", locs, 4), ("conf", confs, C), ("mask", masks, nproto)):
        o = m.add(T(f"{name}_all", (1, npri, last), "u8", scale=m.tensors[parts[0]].scale, zp=128))
        m.ops.append(Op("CONCATENATION", parts, [o], axis=1))
        outs.append(o)
    # output 4: per-cell class logits of the S/8 level, [1, (S/8)^2, C] (src/yolact.rs:91, :108)
    cl, _ = conv(levels[0][0], fpn, C, 1, 1, h8, act=0, so=0.0625, zo=128)
    clq = m.add(T("cells_q", (1, h8, h8, C), "u8", scale=0.0078125, zp=128))
    m.ops.append(Op("QUANTIZE", [cl], [clq]))
    cells = m.add(T("cells", (1, h8 * h8, C), "u8", scale=0.0078125, zp=128))
    m.ops.append(Op("RESHAPE", [clq], [cells], new_shape=[1, h8 * h8, C]))
    m.inputs = [x]
    m.outputs = [outs[0], outs[1], outs[2], proto, cells]
    return m


def random_dag(rng, n_ops=36, hw=(9, 7)):
    """A random branchy uint8 graph for the plan-level transformations (operator fusion, depth ordering, grouped launches): every
    new operator reads tensors picked at random among those that exist, so independent convolutions of equal shape at one depth,
    diamonds, ADDs of branches, CONCATENATIONs whose parts have (sometimes) the output's own quantisation, and chains of
    element-wise operators all occur. Outputs: every tensor nobody reads, plus a few that somebody does."""
    m = Model()
    H, W = hw
    sc = lambda: float(rng.uniform(0.02, 0.08))
    zp = lambda: int(rng.integers(90, 160))
    x = m.add(T("input", (1, H, W, 16), "u8", scale=1 / 128, zp=128))
    avail = [(x, 16)]          # (tensor index, channels); all are [1, H, W, c]
    readers = {x: 0}

    def new(name, c, scale=None, z=None):
        t = m.add(T(f"{name}{len(m.tensors)}", (1, H, W, c), "u8", scale=scale or sc(), zp=zp() if z is None else z))
        avail.append((t, c)); readers[t] = 0
        return t

    def use(*ts):
        for t in ts: readers[t] += 1

    for _ in range(n_ops):
        kind = rng.choice(["conv1", "conv1", "conv3", "conv3", "dw", "add", "cat", "ew"])
        src, c = avail[int(rng.integers(0, len(avail)))]
        if kind in ("conv1", "conv3"):
            k = 1 if kind == "conv1" else 3
            co = int(rng.choice([16, 32, 48]))
            w = m.add(T(f"w{len(m.tensors)}", (co, k, k, c), "u8", data=_q(rng, (co, k, k, c), 96, 160), scale=0.004, zp=128))
            b = m.add(T(f"b{len(m.tensors)}", (co,), "i32", data=rng.integers(-300, 300, co), scale=0.0002, zp=0))
            y = new("c", co)
            m.ops.append(Op("CONV_2D", [src, w, b], [y], padding=0, stride_w=1, stride_h=1, act=int(rng.choice([0, 1, 3]))))
            use(src)
        elif kind == "dw":
            w = m.add(T(f"dw{len(m.tensors)}", (1, 3, 3, c), "u8", data=_q(rng, (1, 3, 3, c), 64, 192), scale=0.01, zp=128))
            b = m.add(T(f"db{len(m.tensors)}", (c,), "i32", data=rng.integers(-300, 300, c), scale=0.0005, zp=0))
            y = new("d", c)
            m.ops.append(Op("DEPTHWISE_CONV_2D", [src, w, b], [y], padding=0, stride_w=1, stride_h=1, act=3, depth_multiplier=1))
            use(src)
        elif kind == "add":
            same = [t for t, cc in avail if cc == c and t != src]
            if not same: continue
            other = same[int(rng.integers(0, len(same)))]
            y = new("a", c)
            m.ops.append(Op("ADD", [src, other], [y], act=int(rng.choice([0, 1]))))
            use(src, other)
        elif kind == "cat":
            parts = list(dict.fromkeys([src] + [avail[int(rng.integers(0, len(avail)))][0] for _ in range(int(rng.integers(1, 3)))]))
            if len(parts) < 2: continue
            ctot = sum(m.tensors[t].shape[3] for t in parts)
            if rng.random() < 0.5:   # the output takes its first part's quantisation (that part can then be written in place)
                y = new("k", ctot, scale=m.tensors[parts[0]].scale, z=m.tensors[parts[0]].zp)
            else:
                y = new("k", ctot)
            m.ops.append(Op("CONCATENATION", list(parts), [y], axis=3))
            use(*parts)
        else:
            code = str(rng.choice(["RELU", "QUANTIZE", "TANH"]))
            if code == "TANH":
                y = new("t", c, scale=1 / 128, z=128)
            else:
                y = new("e", c)
            m.ops.append(Op(code, [src], [y]))
            use(src)
    leaves = [t for t, _ in avail if readers[t] == 0 and t != x]
    inner = [t for t, _ in avail if readers[t] > 0 and t != x]
    extra = [inner[int(i)] for i in rng.integers(0, len(inner), min(3, len(inner)))] if inner else []
    m.inputs = [x]
    m.outputs = list(dict.fromkeys(leaves + extra))
    return m
