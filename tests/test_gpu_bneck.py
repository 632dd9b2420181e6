"""-m gpu: bottleneck chains (csrc/bneck.hip, yh_tuning.chain): an identity block's 3x3 conv, its last 1x1 conv with the residual
add and the next block's first 1x1 conv as ONE launch - inside interpreter.invoke() (/root/reference/src/yolact.rs:163) these
are CONV_2D, CONV_2D + ADD, CONV_2D of the op histogram (data/FRC_model_edgetpu.log:7-19).

The fused launch is bit-transparent by construction (same MFMA products in the same order, same f32 epilogue operations,
intermediates rounded where the separate launches store them), so the bar is exact equality with the unfused engine
(chain = 0) on every tensor both write, on the heads and on the detections - for every tile variant."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
S = 550


def _pair(ya, n, **tune):
    f = ya.Engine(input_size=S, max_batch=n, use_graph=True, tune=dict(tune) or None)
    u = ya.Engine(input_size=S, max_batch=n, use_graph=True, tune=dict(tune, chain=0))
    blob = f.generate_weights(seed=1)
    f.load_weights(blob); u.load_weights(blob)
    return f, u


DUAL = "bneck_chain_f16<64,128,next,dual"


XN = "bneck_xn_f16"   # (round 4's 128-pixel and pipelined forms of this launch are retired: tools/study/retired_r05_forms.patch)


@pytest.mark.parametrize("n,tune,want,chains,nexts,xn", [
    (1, {}, {"bneck_chain_f16<64,64", "bneck_chain_f16<128,64", DUAL}, 6, 4, 0),                         # batch 1: the small tiles
    (8, {}, {"bneck_chain_f16<64,64", "bneck_chain_f16<128,64", DUAL, XN}, 6, 4, 4),                     # batch 8: ... and layer 3's expand + next-reduce launches (154 tiles: more than half a round, at most one)
    (8, {"chain": 17 + 128}, {"bneck_chain_f16<64,64", "bneck_chain_f16<128,64", DUAL}, 6, 4, 0),        # ... without them (bit 7)
    (8, {"plan_cus": 64}, {"bneck_chain_f16<64,256", "bneck_chain_f16<128,64", DUAL}, 6, 4, 0),          # planned for a 64-CU chip: 595 big tiles in layer 1 (>= 8 per CU), 298 in layer 2
    (8, {"plan_cus": 32}, {"bneck_chain_f16<64,256", "bneck_chain_f16<128,128", DUAL}, 6, 4, 0),         # planned for a 32-CU chip: both big tiles + the first-block form (the default plan of a batch-64 step)
    (8, {"plan_cus": 32, "chain": 65}, {"bneck_chain_f16<64,256", "bneck_chain_f16<128,128"}, 5, 3, 0),  # without the first-block form (bit 6)
    (2, {"chain": 17 + 2}, {XN}, 6, 4, 4),                                                               # the 64-pixel form forced outside its window (bit 1), 38 tiles + 18 rows
])
def test_chain_equals_separate_launches_bit_for_bit(built, n, tune, want, chains, nexts, xn):
    import yolact_amd as ya
    f, u = _pair(ya, n, **tune)
    rng = np.random.default_rng(100 + n)
    for rep in range(2):
        frames = rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8)
        for e in (f, u):
            e.set_input(frames); e.evaluate()
        names = [p["name"] for p in f.profile(with_tail=True, reps=1)]
        got = {nm.split(":")[0].replace(",next>", "").rstrip(">") for nm in names if nm.startswith("bneck_")}
        assert {w for w in want} <= got, (want, got)
        # the chains of R50's layers 1-2: l1b0 (first block, + l1b1_a), l1b1 (+ l1b2_a), l1b2, l2b1 (+ l2b2_a), l2b2 (+ l2b3_a), l2b3
        # ... and of layer 3 (256 planes: the 3x3 conv stays a launch of its own): l3b1 (+ l3b2_a) ... l3b4 (+ l3b5_a)
        n_xn = sum(nm.startswith("bneck_xn") for nm in names)
        assert n_xn == xn, names
        assert sum(nm.startswith("bneck_chain") for nm in names) == chains and sum(",next" in nm for nm in names) == nexts, names
        assert len(names) == len(u.profile(with_tail=True, reps=1)) - chains - nexts - xn
        for name in ("l1b0", "l1b1_a", "l1b1", "c2", "l1b2_a", "l2b1", "l2b2_a", "l2b3_a", "c3", "l3b1", "l3b2_a", "l3b4", "l3b5_a", "c4", "c5", "p3", "proto2"):
            for fr in (0, n - 1):
                assert np.array_equal(f.tensor_frame(name, fr), u.tensor_frame(name, fr)), (name, fr, rep)
        for i in range(4):
            assert np.array_equal(f.output(i), u.output(i)), i
        for fr in range(n):
            (da, ma), (db, mb) = f.detections(fr), u.detections(fr)
            assert da == db and np.array_equal(ma, mb), fr
    with pytest.raises(ya.YhError):
        f.tensor_frame("l1b1_b", 0)          # the 3x3 conv's output stays in LDS
    assert u.tensor_frame("l1b1_b", 0).shape == (138, 138, 64)
    f.close(); u.close()


def test_chain_layers_against_the_oracle(built, oracle):
    """... and not only against the library's own other path: block outputs of a fused batch-2 run against the oracle's forward."""
    import yolact_amd as ya
    eng = ya.Engine(input_size=S, max_batch=2, use_graph=True, tune=dict(chain=17))   # (batch 2 only fills the 64-pixel tiles: bit 4)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    frames = np.random.default_rng(9).integers(0, 256, (2, S, S, 3), dtype=np.uint8)
    eng.set_input(frames); eng.evaluate()
    assert any(p["name"].startswith("bneck_chain") for p in eng.profile(with_tail=False, reps=1))
    net = oracle.Net(50, S, 81, blob=blob)
    net.forward(frames[1:2], f16=True)
    for name, tol in (("l1b1", 4e-3), ("c2", 6e-3), ("l1b2_a", 6e-3), ("l2b2_a", 1e-2), ("c3", 1e-2)):
        a, b = eng.tensor_frame(name, 1), net.get(name)[0]
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()))
    eng.close()
    # layer 3's expand + next-reduce launch: its two outputs per block against the oracle's forward
    eng = ya.Engine(input_size=S, max_batch=2, use_graph=True, tune=dict(chain=17 + 2))
    eng.load_weights(blob)
    eng.set_input(frames); eng.evaluate()
    assert sum(p["name"].startswith(XN + ":") for p in eng.profile(with_tail=False, reps=1)) == 4
    for name, tol in (("l3b1", 1.5e-2), ("l3b2_a", 1.5e-2), ("l3b4", 1.5e-2), ("l3b5_a", 1.5e-2), ("c4", 1.5e-2)):
        a, b = eng.tensor_frame(name, 1), net.get(name)[0]
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()))
    eng.close()


def test_layer3_launch_in_fp8_precision_equals_separate_launches(built):
    """configs[4]'s own share (YOLACT-700 R101, fp8 precision, 8 frames per GPU) is where layer 3's expand + next-reduce launch runs
    (242 tiles on 256 CUs), and there its second output feeds an fp8 convolution: a' is written as E4M3 codes by the fused launch.
    Same bytes as the separate launches on every head output and detection."""
    import yolact_amd as ya
    n, s = 8, 700
    frames = np.random.default_rng(21).integers(0, 256, (n, s, s, 3), dtype=np.uint8)
    engs, blob = [], None
    for tune in ({}, {"chain": 17 + 128}):
        e = ya.Engine(input_size=s, backbone=101, max_batch=n, use_graph=True, precision=ya.PRECISION_FP8, tune=tune or None)
        if blob is None:
            blob = e.generate_weights(seed=1)
        e.load_weights(blob)
        e.set_input(frames)
        e.fp8_calibrate()
        e.evaluate()
        engs.append(e)
    f, u = engs
    names = [p["name"] for p in f.profile(with_tail=True, reps=1)]
    assert sum(nm.startswith(XN + ":") for nm in names) == 21, names          # l3b1 ... l3b21 (+ the next block's reduce conv each)
    assert not any(nm.startswith("bneck_xn") for nm in (p["name"] for p in u.profile(with_tail=True, reps=1)))
    for i in range(4):
        assert np.array_equal(f.output(i), u.output(i)), i
    for fr in range(n):
        (da, ma), (db, mb) = f.detections(fr), u.detections(fr)
        assert da == db and np.array_equal(ma, mb), fr
    f.close(); u.close()
