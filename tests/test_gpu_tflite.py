"""-m gpu: the TFLite model path: reader + quantised executor through the C ABI,
bit for bit against oracle/tfl_oracle.py on synthetic models (the reference's model file is absent)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfl_builder as B
import tfl_models as M

pytestmark = pytest.mark.gpu


def _run_both(model, x, tune=None):
    import tfl_oracle as O
    import yolact_amd as ya
    eng = ya.TfliteEngine(B.serialize(model), tune=tune)
    eng.set_input(x)
    eng.invoke()
    val = O.run_model(model, {model.inputs[0]: x})
    outs = [eng.output(i) for i in range(eng.output_count())]
    return eng, val, outs


CASES = [("CONV_2D", dict()), ("CONV_2D", dict(k=1, stride=1, act=1, ci=16, co=5)), ("CONV_2D", dict(k=3, stride=2, padding=0, act=3, h=11, w=10)),
         ("CONV_2D", dict(k=3, stride=2, padding=1, h=12, w=9)), ("CONV_2D", dict(k=5, stride=1, padding=1, h=9, w=9, so=0.3)),
         ("DEPTHWISE_CONV_2D", dict()), ("DEPTHWISE_CONV_2D", dict(stride=2, act=3, h=10, w=11, ci=16)), ("DEPTHWISE_CONV_2D", dict(dm=2, ci=4)),
         ("ADD", dict()), ("ADD", dict(act=1)), ("PAD", dict()), ("RESIZE_BILINEAR", dict()), ("RESIZE_BILINEAR", dict(hi=7, wi=7, ho=14, wo=14, align_corners=True)),
         ("RESIZE_BILINEAR", dict(hi=4, wi=6, ho=7, wo=9, half_pixel_centers=True)), ("TANH", dict()), ("RELU", dict()), ("QUANTIZE", dict()),
         ("CONCATENATION", dict()), ("CONCATENATION", dict(axis=-1)), ("RESHAPE", dict()), ("DEQUANTIZE", dict())]


@pytest.mark.parametrize("code,kw", CASES)
def test_single_op_bit_exact(built, code, kw):
    rng = np.random.default_rng(hash(code) % 1000 + len(kw))
    model = M.single_op(code, rng, **kw)
    x = rng.integers(0, 256, model.tensors[model.inputs[0]].shape, dtype=np.uint8)
    eng, val, outs = _run_both(model, x)
    want = val[model.outputs[0]]
    assert outs[0].shape == want.shape and outs[0].dtype == want.dtype
    assert np.array_equal(outs[0], want)
    eng.close()


def test_known_answers_dequantize_and_saturation(built):
    """SURVEY Appendix A.5 through the model path; conv saturation at both clamp ends."""
    rng = np.random.default_rng(0)
    m = M.single_op("DEQUANTIZE", rng)
    x = np.zeros(m.tensors[0].shape, np.uint8)
    x.flat[0], x.flat[1] = 130, 0
    m.tensors[0].scale, m.tensors[0].zp = 0.5, 128
    eng, val, outs = _run_both(m, x)
    assert outs[0].flat[0] == 1.0 and outs[0].flat[1] == -64.0
    eng.close()
    m = M.single_op("CONV_2D", rng, so=0.0005)      # tiny output scale: everything saturates
    x = rng.integers(0, 256, m.tensors[0].shape, dtype=np.uint8)
    eng, val, outs = _run_both(m, x)
    assert np.array_equal(outs[0], val[m.outputs[0]])
    assert np.isin(outs[0], (0, 255)).mean() > 0.9     # both clamp ends are exercised
    eng.close()


def _surviving(eng, model, val, ya):
    """Every tensor the plan still writes == the oracle; returns how many are folded away (reading one says so: YH_ESTATE)."""
    gone = 0
    for i in sorted({o for op in model.ops for o in op.outputs}):
        t = model.tensors[i]
        try:
            got = eng.tensor(i, t.shape, B.NP_TYPE[t.dtype])
        except ya.YhError as e:
            assert e.code == ya.capi.ESTATE and "folded" in str(e), (i, t.name, str(e))
            gone += 1
            continue
        assert np.array_equal(got, val[i]), (i, t.name)
    return gone


def test_operator_fusion_keeps_every_surviving_tensor_bit_exact(built):
    """yh_tuning.tfl_fuse (default 1): QUANTIZE / RELU / RELU6 / TANH / ADD folded into the epilogue of the
    convolution or resize that produces their operand, PAD into the convolution behind it, CONCATENATION parts written in place.
    Bit-exactness against the numpy oracle on every tensor the fused plan still writes (the 40-op graph: conv -> ADD -> RELU,
    resize -> nothing (its ADD's other producer runs later), conv -> TANH with two readers, conv -> QUANTIZE -> RESHAPE -> output;
    the 136-op stand-in: 9 residual ADDs, 2 FPN ADDs, 2 PADs, 5 TANH + 16 QUANTIZE chains into 3 CONCATENATIONs) and equality with
    the unfused plan (tfl_fuse = 0, the checker) on every output; launch counts from yh_tfl_plan_info. The fused engine also runs
    yh_tuning.tfl_group (default): the plan in depth order with the independent register-fed convolutions of one kernel form as one
    launch (the head's convolutions over the five pyramid levels) - the checker does neither, so equality here covers both."""
    import yolact_amd as ya
    rng = np.random.default_rng(7)
    small = M.mobilenet_like(rng, S=64, C=6)
    x = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    eng, val, outs = _run_both(small, x)
    gone = _surviving(eng, small, val, ya)
    assert gone >= 4, gone                                # padded, p3sum, the lateral conv's own output, conf / conf_q ...
    for k, o in enumerate(small.outputs):
        assert np.array_equal(outs[k], val[o]), small.tensors[o].name
    eng.close()
    rng = np.random.default_rng(31)
    big = M.mobilenetv2_yolact(rng)
    blob = bytes(B.serialize(big))
    x = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)
    import tfl_oracle as TO
    vals = [TO.run_model(big, {big.inputs[0]: x[i:i + 1]}) for i in range(2)]
    fused, plain = ya.TfliteEngine(blob), ya.TfliteEngine(blob, tune=dict(tfl_fuse=0, tfl_group=0))   # (plain: one launch per operator, in file order)
    pf, pp = fused.plan_summary(), plain.plan_summary()
    print("fused plan:", pf, " unfused:", pp)
    assert pp["launches_per_invoke"] >= 130 and pf["launches_per_invoke"] <= 60, (pf, pp)   # (82 after operator fusion, 58 with the independent convolutions grouped)
    assert pf["conv2d_launches"] == pp["conv2d_launches"] == 64
    for e in (fused, plain):
        e.set_batch(2); e.set_input(x); e.invoke()
    for k, o in enumerate(big.outputs):
        a, b = fused.output(k), plain.output(k)
        assert np.array_equal(a, b), big.tensors[o].name
        for i in range(2):
            assert np.array_equal(a[i], np.asarray(vals[i][o]).reshape(a[i].shape)), (big.tensors[o].name, i)
    fused.set_batch(1); fused.set_input(x[:1]); fused.invoke()
    gone = _surviving(fused, big, vals[0], ya)
    assert gone >= 45, gone                               # 9 + 2 ADD operands, 2 padded, 21 TANH / QUANTIZE inputs, 15 concat parts ...
    frame = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
    fa, fb = frame.copy(), frame.copy()
    fused.classify_frame(fa, 640, 480, ya.COMPAT_SANE)
    plain.classify_frame(fb, 640, 480, ya.COMPAT_SANE)
    assert np.array_equal(fa, fb)
    fused.close(); plain.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_branchy_graphs_fused_grouped_and_plain_plans_agree(built, seed):
    """Random DAGs (tests/tfl_models.random_dag: independent convolutions of equal shape at one depth, diamonds, ADDs of branches,
    CONCATENATIONs with and without in-place parts, element-wise chains): the default plan - operators folded into their producers,
    depth order, independent register-fed convolutions of one form as one launch - against the numpy oracle on every output and
    every tensor the plan still writes, and against the plain plan (one launch per operator, file order) on every output; one and
    two images per invoke. A wrong dependency in the reordering, a group member launched before its producer or a folded
    operator applied twice would show here."""
    import tfl_oracle as O
    import yolact_amd as ya
    rng = np.random.default_rng(1000 + seed)
    model = M.random_dag(rng, n_ops=36 + 2 * seed)
    blob = bytes(B.serialize(model))
    x = rng.integers(0, 256, (2, 9, 7, 16), dtype=np.uint8)
    vals = [O.run_model(model, {model.inputs[0]: x[i:i + 1]}) for i in range(2)]
    fast, plain = ya.TfliteEngine(blob), ya.TfliteEngine(blob, tune=dict(tfl_fuse=0, tfl_group=0))
    nogroup = ya.TfliteEngine(blob, tune=dict(tfl_group=0))
    lf, lp, ln = (e.plan_summary()["launches_per_invoke"] for e in (fast, plain, nogroup))
    assert lf <= ln <= lp, (lf, ln, lp)
    for nb in (2, 1):
        for e in (fast, plain):
            e.set_batch(nb); e.set_input(x[:nb]); e.invoke()
        for k, o in enumerate(model.outputs):
            a, b = fast.output(k), plain.output(k)
            assert np.array_equal(a, b), (seed, nb, model.tensors[o].name)
            for i in range(nb):
                assert np.array_equal(np.asarray(a).reshape((nb,) + tuple(np.asarray(vals[i][o]).shape[1:]))[i], np.asarray(vals[i][o])[0]), (seed, nb, model.tensors[o].name, i)
    _surviving(fast, model, vals[0], ya)    # (batch 1 was the last invoke)
    for e in (fast, plain, nogroup): e.close()
    print(f"seed {seed}: {lp} launches plain, {ln} fused, {lf} fused + grouped")


def test_mobilenet_like_graph_every_tensor(built):
    """Every activation of a 40-op MobileNetV2-FPN-shaped graph, not only the outputs (one launch per operator: tfl_fuse = 0)."""
    rng = np.random.default_rng(7)
    model = M.mobilenet_like(rng, S=64, C=6)
    x = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    eng, val, outs = _run_both(model, x, tune=dict(tfl_fuse=0))
    assert eng.output_count() == 5
    info = eng.output_info(4)
    assert info["dims"] == (1, 64, 6) and info["kind"] == 3 and info["scale"] == 0.0078125 and info["zero_point"] == 128
    written = {o for op in model.ops for o in op.outputs}
    for i in sorted(written):
        t = model.tensors[i]
        got = eng.tensor(i, t.shape, B.NP_TYPE[t.dtype])
        assert np.array_equal(got, val[i]), (i, t.name)
    for k, o in enumerate(model.outputs):
        assert np.array_equal(outs[k], val[o])
    assert len(np.unique(outs[4])) > 8               # not a degenerate constant output
    eng.close()


@pytest.mark.parametrize("seed", [11, 0])
def test_full_size_mobilenetv2_yolact_graph(built, seed):
    """The full-size stand-in for FRC_model.tflite (136 ops at 224x224: the op census of
    data/FRC_model_edgetpu.log): all five outputs and a sample of intermediates against the numpy
    oracle, bit for bit, and a second invoke (graph replay) identical to the first."""
    rng = np.random.default_rng(seed)
    model = M.mobilenetv2_yolact(rng)
    x = rng.integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
    eng, val, outs = _run_both(model, x, tune=dict(tfl_fuse=seed != 0))   # (one seed per plan: fused / one launch per operator)
    assert eng.output_count() == 5 and eng.output_info(4)["dims"] == (1, 784, 81)
    for k, o in enumerate(model.outputs):
        assert np.array_equal(outs[k], val[o]), model.tensors[o].name
    import yolact_amd as ya
    written = sorted({o for op in model.ops for o in op.outputs})
    for i in written[::7]:
        t = model.tensors[i]
        try:
            assert np.array_equal(eng.tensor(i, t.shape, B.NP_TYPE[t.dtype]), val[i]), (i, t.name)
        except ya.YhError as e:
            assert seed != 0 and e.code == ya.capi.ESTATE   # folded into its producer's launch (fused plan only)
    eng.invoke()
    for k, o in enumerate(model.outputs):
        assert np.array_equal(eng.output(k), outs[k])
    eng.close()


def test_graph_replay_equals_eager_over_200_replays(built):
    """yh_tuning.tfl_graph = 1 (the plan as one captured hipGraph, with the one-node second branch every capture of this
    library carries: DESIGN.md §8): 200 replays on changing inputs, every output equal to the eager executor's bit for bit."""
    import yolact_amd as ya
    rng = np.random.default_rng(5)
    model = M.mobilenetv2_yolact(rng)
    buf = bytes(B.serialize(model))
    eager, graph = ya.TfliteEngine(buf, tune=dict(tfl_graph=0)), ya.TfliteEngine(buf, tune=dict(tfl_graph=1))
    for it in range(200):
        x = rng.integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
        for e in (eager, graph):
            e.set_input(x)
            e.invoke()
        if it % 20 == 0 or it == 199:
            for k in range(5):
                assert np.array_equal(eager.output(k), graph.output(k)), (it, k)
    frame = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
    fa, fb = frame.copy(), frame.copy()
    eager.classify_frame(fa, 640, 480, ya.COMPAT_SANE)
    graph.classify_frame(fb, 640, 480, ya.COMPAT_SANE)
    assert np.array_equal(fa, fb)
    eager.close(); graph.close()


def test_classify_through_a_tflite_model(built, oracle, golden_dir):
    """Yolact::classify (yolact.rs:192-234) with a .tflite in the middle: pre-processing, two
    invokes, output-4 dequantisation (yolact.rs:177), postprocess, stitch, resize back."""
    from PIL import Image
    import tfl_oracle as O
    import yolact_amd as ya
    rng = np.random.default_rng(11)
    S, C = 64, 6
    model = M.mobilenet_like(rng, S=S, C=C)
    eng = ya.TfliteEngine(B.serialize(model))
    rgb = np.asarray(Image.open(os.path.join(golden_dir, "frc_balls.png")).convert("RGB").resize((160, 120), Image.BILINEAR))
    frame0 = oracle.pack_rgb(rgb)
    tiles = oracle.classify_pre(frame0, 160, 120, S)
    cells = []
    for t in range(2):
        val = O.run_model(model, {model.inputs[0]: tiles[t:t + 1]})
        q = val[model.outputs[4]]
        cells.append(oracle.dequant_u8(q.reshape(-1), 0.0078125, 128).reshape(S // 8 * (S // 8), C))
    cells = np.stack(cells)
    for mode in (ya.COMPAT_SANE, ya.COMPAT_STRICT):
        rc, want = oracle.classify_post(cells, S, C, mode, 160, 120)
        frame = frame0.copy()
        if rc:
            with pytest.raises(ya.YhError) as e:
                eng.classify_frame(frame, 160, 120, mode)
            assert e.value.code == ya.EDIVERGE and np.array_equal(frame, frame0)
        else:
            eng.classify_frame(frame, 160, 120, mode)
            assert np.array_equal(frame, want)
    eng.close()


def test_unsupported_models_are_rejected_with_a_reason(built):
    import yolact_amd as ya
    rng = np.random.default_rng(3)
    m = M.single_op("CONV_2D", rng)
    m.tensors[1].scale = None                          # weights without quantisation parameters
    with pytest.raises(ya.YhError) as e:
        ya.TfliteEngine(B.serialize(m))
    assert "uint8" in str(e.value)
    with pytest.raises(ya.YhError):
        ya.TfliteEngine(b"not a flatbuffer at all.....")
    m = M.single_op("ADD", rng)
    eng = ya.TfliteEngine(B.serialize(m))
    with pytest.raises(ya.YhError):
        eng.set_input(np.zeros(3, np.uint8))           # wrong size is an error here (the reference only warns, yolact.rs:151-158)
    eng.close()


def test_yolact_init_from_model_file(built, oracle, tmp_path, golden_dir):
    """`Yolact::init()` with a model file on disk, as the reference loads one (yolact.rs:18-20)."""
    from PIL import Image
    import yolact_amd as ya
    rng = np.random.default_rng(5)
    path = tmp_path / "FRC_model.tflite"
    path.write_bytes(B.serialize(M.mobilenet_like(rng, S=64, C=6)))
    y = ya.Yolact.init(model_path=str(path), compat_mode=ya.COMPAT_SANE)
    rgb = np.asarray(Image.open(os.path.join(golden_dir, "red_robot.png")).convert("RGB").resize((640, 480), Image.BILINEAR))
    buf = oracle.pack_rgb(rgb)
    y.classify(buf)
    assert ((buf >> 24) <= 3).all() and (buf & 0xFFFF == 0).all()
    with pytest.raises(FileNotFoundError):
        ya.Yolact.init(model_path=str(tmp_path / "missing.tflite"))
    y.interpreter.close()


def test_conv_random_geometries_bit_exact(built):
    """Seeded sweep over CONV_2D / DEPTHWISE_CONV_2D geometries: input channels that take the dot-product
    kernel with the K split over waves (Ci % 16 == 0), without it (Ci % 4 == 0) and the scalar kernel
    (other Ci), output channels around the 8-wide channel block, kernel 1/3/5, stride 1/2, SAME / VALID,
    every fused activation - each against the numpy oracle, bit for bit."""
    rng = np.random.default_rng(99)
    for _ in range(60):
        code = "CONV_2D" if rng.random() < 0.75 else "DEPTHWISE_CONV_2D"
        k, stride = int(rng.choice([1, 3, 5])), int(rng.choice([1, 2]))
        h, w = int(rng.integers(k, 20)), int(rng.integers(k, 20))
        kw = dict(k=k, stride=stride, padding=int(rng.integers(0, 2)), act=int(rng.choice([0, 1, 3])), h=h, w=w,
                  ci=int(rng.choice([3, 4, 8, 12, 16, 32, 48, 64, 96])), co=int(rng.choice([1, 5, 8, 9, 16, 24, 31, 64])),
                  so=float(rng.uniform(0.02, 0.3)))
        if code == "DEPTHWISE_CONV_2D":
            kw["dm"] = int(rng.choice([1, 1, 2]))
        model = M.single_op(code, rng, **kw)
        t_in = model.tensors[model.inputs[0]]
        x = rng.integers(0, 256, t_in.shape, dtype=np.uint8)
        eng, val, outs = _run_both(model, x)
        assert np.array_equal(outs[0], val[model.outputs[0]]), (code, kw)
        eng.close()


@pytest.mark.parametrize("graph", [0, 1])
def test_batch_plan_two_tiles_in_one_invoke_equals_two_invokes(built, graph):
    """yh_tfl_set_batch(2): the reference invokes its batch-1 model once per tile (yolact.rs:216-217); the two tiles are
    independent, so they run as ONE pass of the plan with image-major activations. Every written tensor of the full-size
    stand-in for both images == the numpy oracle on each image alone, bit for bit; then back to one image per invoke."""
    import yolact_amd as ya
    rng = np.random.default_rng(23)
    model = M.mobilenetv2_yolact(rng)
    x2 = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)
    eng = ya.TfliteEngine(B.serialize(model), tune=dict(tfl_graph=graph, tfl_fuse=0))
    eng.set_batch(2)
    eng.set_input(x2)
    eng.invoke()
    import tfl_oracle as TO
    vals = [TO.run_model(model, {model.inputs[0]: x2[i:i + 1]}) for i in range(2)]
    written = sorted({o for op in model.ops for o in op.outputs})
    for i in written[::5] + list(model.outputs):
        t = model.tensors[i]
        got = eng.tensor(i, t.shape, B.NP_TYPE[t.dtype])
        for k in range(2):
            assert np.array_equal(got[k], np.asarray(vals[k][i]).reshape(got[k].shape)), (i, t.name, k)
    for rep in range(3):                                  # replay: the same bits
        eng.invoke()
        assert np.array_equal(eng.output(4)[1], np.asarray(vals[1][model.outputs[4]]).reshape(eng.output(4)[1].shape))
    eng.set_batch(1)
    eng.set_input(x2[1:2])
    eng.invoke()
    assert np.array_equal(eng.output(4), vals[1][model.outputs[4]])
    with pytest.raises(ya.YhError):
        eng.set_batch(3)
    eng.close()


def test_int8_mfma_conv_geometries_bit_exact(built):
    """CONV_2D with Ci % 4 == 0 runs on v_mfma_i32_16x16x64_i8 (yh_tuning.tfl_dot = 3, the default: one wave per 16 x 16 tile,
    operands loaded straight into the MFMA's registers, dwords past Ci fed zeros; tfl_dot = 2: 64 x 64 tiles through LDS, Ci %
    64 == 0 - other layers on the v_dot4 kernel): uint8 operands flipped to int8, sum(x'w') in int32, TFLite's value restored exactly from sum x' (v_dot4 beside the MFMAs), sum w' (tabulated) and the
    zero points; a padded tap is fed the zero point. Seeded sweep: kernel 1/3/5, stride 1/2, SAME/VALID, every fused activation,
    output channels around the 64-channel tile and not a multiple of 4 (243: byte stores), pixel counts around the 64-pixel
    tile, one and two images per invoke, input channels 4 ... 192 (the tail chunk of 4, 8, 16, 24, 32, 48, 96 and 144 channels) - against
    the numpy oracle under every tfl_dot, bit for bit."""
    import tfl_oracle as O
    import yolact_amd as ya
    rng = np.random.default_rng(2024)
    for it in range(48):
        k, stride = int(rng.choice([1, 3, 3, 5])), int(rng.choice([1, 1, 2]))
        h, w = int(rng.integers(k, 24)), int(rng.integers(k, 24))
        kw = dict(k=k, stride=stride, padding=int(rng.integers(0, 2)), act=int(rng.choice([0, 1, 3])), h=h, w=w,
                  ci=int(rng.choice([64, 128, 192] if it < 24 else [4, 8, 16, 24, 32, 48, 96, 144, 160])), co=int(rng.choice([1, 12, 31, 33, 63, 64, 65, 96, 128, 243])), so=float(rng.uniform(0.05, 0.6)))
        model = M.single_op("CONV_2D", rng, **kw)
        t_in = model.tensors[model.inputs[0]]
        blob = B.serialize(model)
        nb = 1 + it % 2
        x = rng.integers(0, 256, (nb,) + tuple(t_in.shape[1:]), dtype=np.uint8)
        want = [O.run_model(model, {model.inputs[0]: x[i:i + 1]})[model.outputs[0]] for i in range(nb)]
        for dot in (3, 2, 1):
            eng = ya.TfliteEngine(blob, tune=dict(tfl_dot=dot))
            eng.set_batch(nb)
            eng.set_input(x)
            eng.invoke()
            got = eng.output(0).reshape((nb,) + tuple(np.asarray(want[0]).shape[1:]))
            for i in range(nb):
                assert np.array_equal(got[i], np.asarray(want[i])[0]), (kw, nb, dot, i)
            eng.close()
