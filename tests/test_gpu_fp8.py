"""-m gpu: the fp8 forward (yh_config.precision = YH_PRECISION_FP8; BASELINE.json configs[4]).
The K-heavy 3x3 convolutions read OCP E4M3 operands on the block-scaled fp8 MFMA, everything else stays f16.
Parity = HIP engine vs the oracle's fp8 mode (oracle/orc_net.c: the same convolutions named, the same per-input-channel
activation scales folded into the weights, per-output-channel weight scales derived the same way): heads within a stated tolerance, tail bit-exact on
the engine's own heads. The fp8-vs-f16 gap is a property of the configuration and is reported with its own bound.
The reference's model is quantised end to end too (uint8, data/README.md:5-10); nothing in it pins E4M3: parity unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TH = 0.005


def _expand(layers):
    """Engine layer names -> oracle conv names (the merged shared-head launch covers the oracle's five per-level convs)."""
    out = {}
    for name, sc in layers:
        for n in ([f"{name}{l}" for l in range(5)] if name in ("head_t", "head_out") else [name]):
            out[n] = sc
    return out


@pytest.fixture(scope="module")
def fp8_setup(built, oracle):
    import yolact_amd as ya
    S = 160
    eng = ya.Engine(input_size=S, max_batch=2, use_graph=True, precision=ya.PRECISION_FP8, conf_thresh=TH)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    img = np.random.default_rng(8).integers(0, 256, (2, S, S, 3), dtype=np.uint8)
    yield eng, blob, img, S
    eng.close()


def test_fp8_needs_scales_and_lists_its_layers(fp8_setup):
    import yolact_amd as ya
    eng, _, img, _ = fp8_setup
    eng.set_input(img)
    with pytest.raises(ya.YhError) as e:
        eng.evaluate()                                   # no activation scales yet
    assert e.value.code == ya.capi.ESTATE
    names = [n for n, _ in eng.fp8_layers()]
    # R50: the 3x3 convs of layers 3 and 4 (256 / 512 input channels), FPN pred + down, protonet, head trunk
    for want in ("l3b0_b", "l3b5_b", "l4b0_b", "l4b2_b", "p3", "p4", "p5", "p6", "p7", "proto0", "proto1", "proto2", "proto3", "head_t"):
        assert want in names, want
    for never in ("l2b0_b", "l1b0_b", "head_out", "proto", "lat3", "l3b0_a"):
        assert never not in names, never                 # < 256 input channels, 1x1, or 351 output channels: f16
    eng.fp8_calibrate()
    sc = dict(eng.fp8_channel_scales())
    assert all(v.dtype == np.float32 and ((0.0 < v) & (v < 1e3)).all() for v in sc.values())
    assert sc["l3b0_b"].shape == (256,) and sc["l4b1_b"].shape == (512,) and sc["head_t"].shape == (256,)
    # scales belong to the TENSOR: P3..P7 live in one pyramid buffer, so its readers share one vector of channel scales
    assert np.array_equal(sc["p6"], sc["head_t"]) and np.array_equal(sc["p6"], sc["proto0"])
    assert len(np.unique(sc["p3"])) > 16                  # round 4: one scale per input CHANNEL, not per tensor
    assert all(np.float32(v) == sc[n].max() for n, v in eng.fp8_layers())   # (yh_fp8_layer_info reports the largest)


def test_fp8_forward_vs_fp8_oracle(fp8_setup, oracle):
    """Tolerance, stated: both sides quantise the same f16-rounded tensors with the same scales, but their inputs differ
    by f16 summation-order noise, and an activation that sits near an E4M3 rounding boundary then takes the neighbouring
    code on one side (a 6-12 % step of that one operand): a 0.1 % input difference flips 1-2 % of the codes of a tensor, i.e.
    ~1 % rms per quantised layer, and ~13 chained quantised layers compound it. Heads: max |err| <= 15 % of the tensor's
    absmax (25 % for the tanh mask coefficients), rms err <= 10 % of its rms (measured: 4.1 % loc, 6.8 % proto; the f16
    path's bounds are 3 % / 0.5 %; fp8 vs f16 itself differs by 4-10 % rms); the measured figures are printed (pytest -s). Because that end-to-end bound is loose by
    nature, test_fp8_single_layers_are_tight checks every kind of fp8 layer on IDENTICAL quantised inputs, where only the
    summation order remains."""
    eng, blob, img, S = fp8_setup
    eng.set_input(img)
    eng.fp8_calibrate()
    eng.evaluate()
    got = [eng.output(i) for i in range(4)]
    net = oracle.Net(50, S, 81, blob=blob)
    net.set_fp8(_expand(eng.fp8_channel_scales()))
    want = net.forward(img, f16=True)
    sc = dict(eng.fp8_channel_scales())
    consumer = {"l3b0_a": "l3b0_b", "p5": "head_t", "p3": "head_t", "p7": "head_t", "proto1": "proto2"}
    # Intermediate tensors. A tensor that only fp8 convolutions read exists in the engine only as E4M3 codes (what
    # yh_debug_read_tensor returns is their decoded value), while the oracle stores it in f16 and quantises where it
    # is consumed: for those the bound includes E4M3's half step (2^-4 relative = 6.25 %, of at most the tensor's absmax).
    for name, tol, q_only in (("l3b0_a", 2e-2, True), ("l3b0_b", 4e-2, False), ("c4", 4e-2, False), ("c5", 5e-2, False), ("lat5", 5e-2, False),
                              ("p5", 6e-2, True), ("p3", 6e-2, True), ("p7", 8e-2, True), ("proto1", 8e-2, True), ("proto2", 8e-2, False),
                              ("proto3", 0.12, False), ("head_t0", 8e-2, False), ("head_t2", 8e-2, False), ("head_t4", 8e-2, False)):
        a, b = eng.tensor(name), net.get(name)
        assert a.shape == b.shape, name
        bound = (tol + (0.0625 if q_only else 0.0)) * max(1.0, np.abs(b).max())
        print(f"fp8 engine vs fp8 oracle {name}: max |err| {np.abs(a - b).max():.4f} = {100 * np.abs(a - b).max() / max(1.0, np.abs(b).max()):.2f} % of absmax (bound {100 * bound / max(1.0, np.abs(b).max()):.1f} %)")
        assert np.abs(a - b).max() <= bound, (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
        if q_only:   # and every decoded value is an E4M3 value times its channel's scale
            s_t = sc[consumer[name]].astype(np.float64)
            table = np.sort(oracle.e4m3_decode_table()[np.isfinite(oracle.e4m3_decode_table())].astype(np.float64))
            q = (a.astype(np.float64) / s_t).ravel()
            near = table[np.clip(np.searchsorted(table, q), 1, len(table) - 1)]
            near2 = table[np.clip(np.searchsorted(table, q), 1, len(table) - 1) - 1]
            err = np.minimum(np.abs(q - near), np.abs(q - near2))
            assert (err <= 1e-5 * np.maximum(1.0, np.abs(q))).all(), name
    for name, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        mx, rms = float(np.abs(a - b).max() / max(1.0, np.abs(b).max())), float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
        print(f"fp8 engine vs fp8 oracle {name}: max |err| {100 * mx:.2f} % of absmax, rms err {100 * rms:.3f} % of rms")
        assert mx <= (0.25 if name == "mask" else 0.15) and rms <= 0.10, (name, mx, rms)   # (mask = tanh: bounded by 1, a flipped code near 0 moves it most)
    # the tail is precision-independent: bit-exact on the engine's own heads
    pri = net.priors()
    for f in range(2):
        dets, masks = eng.detections(f)
        odets, omasks = oracle.detect(got[0][f], got[1][f], got[2][f], got[3][f], pri, conf_thresh=TH)
        assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets]
        assert np.array_equal(masks, omasks)
    # determinism under graph replay
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), got[i])


def test_fp8_single_layers_are_tight(fp8_setup, oracle):
    """One fp8 convolution at a time on identical inputs: the engine's own E4M3 input tensor (decoded exactly), the blob's
    weights with the input tensor's channel scales folded in along K and then quantised per output channel as DESIGN.md
    §Precision states, f32 convolution of the decoded operands by the oracle, y = relu(fma(acc, s_w, bias)) rounded to f16 -
    against the engine's output of that layer. What differs
    is the summation: 2 f16 ulp + 2^-10 of the sum of |products| (measured: 2^-13 - the block-scaled MFMA adds its 128
    products per instruction with a bounded internal alignment, exact on small integers as tests/test_gpu_ops.py shows). Covers a backbone 3x3 (l3b0_b), a protonet conv (proto1) and the multi-level
    shared head trunk on all five pyramid levels (a tap that left its level would be an O(1) error)."""
    import bench
    eng, blob, img, S = fp8_setup
    eng.set_input(img); eng.fp8_calibrate(); eng.evaluate()
    sc = dict(eng.fp8_channel_scales())
    convs = bench.parse_blob(blob)
    table = oracle.e4m3_decode_table()

    def fq_weights(w, s_c):   # [cout][k][k][cin], channel scales [cin] -> decoded E4M3 values of t = w * s_c and the per-output-channel scale
        t = (w.astype(np.float32) * s_c.astype(np.float32)).astype(np.float32)
        aw = np.abs(t).reshape(t.shape[0], -1).max(1).astype(np.float32)
        sw = np.where(aw > 0, aw / np.float32(448.0), np.float32(1.0)).astype(np.float32)
        inv = (np.float32(1.0) / sw).astype(np.float32)
        q = np.stack([table[oracle.quantize_e4m3(t[o], float(inv[o]))] for o in range(t.shape[0])])
        return q.astype(np.float32), sw

    def check(layer, x_name, y_name, conv_index, stride=1):
        s_x = sc[layer]                                  # one scale per input channel
        xq = table[oracle.quantize_e4m3((eng.tensor(x_name).astype(np.float32) / s_x).astype(np.float32), 1.0)]   # the codes' exact values
        assert np.abs(xq * s_x - eng.tensor(x_name)).max() <= 1e-6 * np.abs(xq * s_x).max()
        w, b = convs[conv_index]
        wq, sw = fq_weights(w, s_x)
        zero = np.zeros(w.shape[0], np.float32)
        acc = oracle.conv2d(xq, wq, zero, stride, 1, None, 0, f16=False)
        mag = oracle.conv2d(np.abs(xq), np.abs(wq), zero, stride, 1, None, 0, f16=False) * sw   # sum of |products|, in output units
        want = np.maximum(acc * sw + b, 0).astype(np.float16).astype(np.float32)
        got = eng.tensor(y_name)
        ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
        err = np.abs(got - want)
        live = want > 0
        ratio = float((err[live] / mag[live]).max())
        print(f"fp8 layer {layer} -> {y_name}: max |err| {err.max():.5f}, {100 * (err > 2 * ulp).mean():.3f} % of elements beyond 2 f16 ulp, max err / sum|products| = 2^{np.log2(max(ratio, 1e-30)):.1f}")
        # the block-scaled MFMA sums 128 products per instruction in hardware: its internal alignment costs up to a few
        # 2^-12 of the sum of |products| on top of the f16 rounding of the result
        assert got.shape == want.shape and (err <= 2 * ulp + 2.0 ** -10 * mag + 1e-6).all(), (layer, y_name, float(err.max()), ratio)
    check("l3b0_b", "l3b0_a", "l3b0_b", 25, stride=2)
    for l in range(5):
        check("head_t", f"p{l + 3}", f"head_t{l}", 66)
    # proto1's output exists only as E4M3 in the engine: compare proto2 (f16 output) from proto1's codes instead
    check("proto2", "proto1", "proto2", 63)


def test_fp8_vs_f16_gap_is_reported_with_its_own_bound(fp8_setup, built):
    """E4M3 has a 3-bit mantissa: against the f16 engine on the same frames the heads move by several per cent rms
    (DESIGN.md §Precision). Stated bound for this configuration: rms difference <= 15 % of the f16 tensor's rms; the
    measured figures are printed (pytest -s) and land in the bench line of configs[4]."""
    import yolact_amd as ya
    eng, blob, img, S = fp8_setup
    ref = ya.Engine(input_size=S, max_batch=2, use_graph=False, conf_thresh=TH)
    ref.load_weights(blob)
    ref.set_input(img); ref.evaluate()
    eng.set_input(img); eng.fp8_calibrate(); eng.evaluate()
    for i, name in enumerate(("loc", "conf", "mask", "proto")):
        a, b = eng.output(i), ref.output(i)
        rel = float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
        print(f"fp8 vs f16 {name}: rms difference {100 * rel:.2f} % of rms")
        assert rel <= 0.15, (name, rel)
    ref.close()


def test_fp8_scales_can_be_set_layer_by_layer(fp8_setup):
    """Stored calibration: setting the same scales by hand reproduces the calibrated forward bit for bit."""
    import yolact_amd as ya
    eng, blob, img, S = fp8_setup
    eng.set_input(img); eng.fp8_calibrate(); eng.evaluate()
    heads = [eng.output(i) for i in range(4)]
    layers = eng.fp8_channel_scales()
    e2 = ya.Engine(input_size=S, max_batch=2, use_graph=False, precision=ya.PRECISION_FP8, conf_thresh=TH)
    for i, (_, sc) in enumerate(layers[:5]):
        e2.fp8_set_layer_scale(i, sc)                     # (scales stored with a model may be set BEFORE its weights are loaded ...)
    e2.load_weights(blob)
    for i, (_, sc) in enumerate(layers):
        if i >= 5:
            e2.fp8_set_layer_scale(i, sc)                 # (... or after)
    e2.set_input(img); e2.evaluate()
    for i in range(4):
        assert np.array_equal(e2.output(i), heads[i])
    with pytest.raises(ya.YhError):
        e2.fp8_set_layer_scale(0, layers[0][1][:8])       # wrong channel count
    e2.close()
    # yh_config.fp8_per_tensor = 1 (round 3's scheme): every channel of a tensor gets the tensor's scale, and ONE number per layer
    # (yh_fp8_set_layer_scale) reproduces that calibrated forward bit for bit
    t1 = ya.Engine(input_size=S, max_batch=2, use_graph=False, precision=ya.PRECISION_FP8, conf_thresh=TH, fp8_per_tensor=True)
    t1.load_weights(blob)
    t1.set_input(img); t1.fp8_calibrate(); t1.evaluate()
    assert all(len(np.unique(v)) == 1 for _, v in t1.fp8_channel_scales())
    assert not np.array_equal(t1.output(1), heads[1])     # (a different quantisation from the per-channel one)
    t2 = ya.Engine(input_size=S, max_batch=2, use_graph=False, precision=ya.PRECISION_FP8, conf_thresh=TH)
    t2.load_weights(blob)
    for i, (_, v) in enumerate(t1.fp8_layers()):
        t2.fp8_set_layer_scale(i, v)
    t2.set_input(img); t2.evaluate()
    for i in range(4):
        assert np.array_equal(t2.output(i), t1.output(i))
    t1.close(); t2.close()
