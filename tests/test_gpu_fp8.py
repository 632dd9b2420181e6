"""-m gpu: the fp8 forward (yh_config.precision = YH_PRECISION_FP8; BASELINE.json configs[4]).
The K-heavy 3x3 convolutions read OCP E4M3 operands on the block-scaled fp8 MFMA, everything else stays f16.
Parity = HIP engine vs the oracle's fp8 mode (oracle/orc_net.c: the same convolutions named, the same per-tensor
activation scales, per-channel weight scales derived the same way): heads within a stated tolerance, tail bit-exact on
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
    sc = dict(eng.fp8_layers())
    assert all(0.0 < v < 1e3 for v in sc.values())
    assert sc["p6"] == sc["head_t"] == sc["proto0"]      # one scale per TENSOR: P3..P7 live in one pyramid buffer


def test_fp8_forward_vs_fp8_oracle(fp8_setup, oracle):
    """Tolerance, stated: both sides quantise the same f16-rounded tensors with the same scales, but their inputs differ
    by f16 summation-order noise, and an activation that sits near an E4M3 rounding boundary then takes the neighbouring
    code on one side (a 6-12 % step of that one operand). Heads: max |err| <= 8 % of the tensor's absmax, rms err <= 2 %
    of its rms (the f16 path's bounds are 3 % / 0.5 %)."""
    eng, blob, img, S = fp8_setup
    eng.set_input(img)
    eng.fp8_calibrate()
    eng.evaluate()
    got = [eng.output(i) for i in range(4)]
    net = oracle.Net(50, S, 81, blob=blob)
    net.set_fp8(_expand(eng.fp8_layers()))
    want = net.forward(img, f16=True)
    for name, tol in (("l3b0_a", 2e-2), ("l3b0_b", 4e-2), ("c4", 4e-2), ("c5", 5e-2), ("lat5", 5e-2), ("p5", 6e-2), ("p3", 6e-2), ("p7", 8e-2), ("proto2", 8e-2),
                      ("proto3", 8e-2), ("head_t0", 8e-2), ("head_t2", 8e-2), ("head_t4", 8e-2)):
        a, b = eng.tensor(name), net.get(name)
        assert a.shape == b.shape, name
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
    for name, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        assert np.abs(a - b).max() <= 0.08 * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
        assert np.sqrt(((a - b) ** 2).mean()) <= 2e-2 * np.sqrt((b ** 2).mean()) + 1e-4, (name, float(np.sqrt(((a - b) ** 2).mean())), float(np.sqrt((b ** 2).mean())))
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
    layers = eng.fp8_layers()
    e2 = ya.Engine(input_size=S, max_batch=2, use_graph=False, precision=ya.PRECISION_FP8, conf_thresh=TH)
    e2.load_weights(blob)
    for i, (_, sc) in enumerate(layers):
        e2.fp8_set_layer_scale(i, sc)
    e2.set_input(img); e2.evaluate()
    for i in range(4):
        assert np.array_equal(e2.output(i), heads[i])
    e2.close()
