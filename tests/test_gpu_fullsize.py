"""-m gpu: parity at BASELINE.json's full size (YOLACT-550 R50, configs[1]/[2]).
One frame against the oracle (the oracle needs a few seconds per 550x550 frame), and exact
size-independent properties on a batch: determinism, frame-permutation equivariance, order and
range invariants of the detection tail."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S = 550


@pytest.fixture(scope="module")
def eng550(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=S, max_batch=8, use_graph=True)
    blob = e.generate_weights(seed=1)
    e.load_weights(blob)
    yield e, blob
    e.close()


def test_geometry_matches_published_yolact550(eng550):
    eng, _ = eng550
    assert eng.P == 19248 and (eng.hp, eng.wp) == (138, 138)
    assert abs(eng.flops_per_frame() / 1e9 - 118.28) < 0.05


def _vs_oracle(eng, net, oracle, img):
    """Engine and oracle on one frame: returns (engine heads, engine dets, oracle heads, oracle dets)."""
    eng.set_input(img)
    eng.evaluate()
    got = [eng.output(i) for i in range(4)]
    dets = eng.detections(0)
    want = net.forward(img, f16=True)
    odets = oracle.detect(want[0][0], want[1][0], want[2][0], want[3][0], net.priors())
    return got, dets, want, odets


def test_one_frame_550_vs_oracle(eng550, oracle, golden_dir):
    """Restated acceptance target at full size: frc_balls.png resized to 550x550, HIP engine vs
    CPU oracle in f16-storage mode: every detection matched by (class id, prior) and mask IoU >= 0.99 computed
    over ALL detections of both sides (per-class union of masks: an unmatched detection counts against it).
    NOT parity with CPU tflite (model file absent). A second, dense frame (uniform noise: 100 detections) has the
    same bar with at most 3 of 100 detections allowed to differ: which of ~130 candidates sits on which side of
    the 0.05 score threshold or the IoU-0.5 NMS test is a discrete decision, and with seeded (untrained) weights a
    few candidates lie within the f16 summation-order noise of those thresholds."""
    import bench
    from PIL import Image
    eng, blob = eng550
    net = oracle.Net(50, S, 81, blob=blob)
    balls = np.asarray(Image.open(os.path.join(golden_dir, "frc_balls.png")).convert("RGB").resize((S, S), Image.BILINEAR))[None]
    noise = np.random.default_rng(5).integers(0, 256, (1, S, S, 3), dtype=np.uint8)
    for name, img, min_dets, max_unmatched in (("frc_balls", balls, 5, 0), ("noise", noise, 90, 3)):
        got, (dets, masks), want, (fdets, fmasks) = _vs_oracle(eng, net, oracle, img)
        for tn, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
            assert np.abs(a - b).max() <= 0.03 * max(1.0, np.abs(b).max()), (name, tn)
            assert np.sqrt(((a - b) ** 2).mean()) <= 5e-3 * np.sqrt((b ** 2).mean()) + 1e-4, (name, tn)
        # tail bit-exact on the engine's own head outputs
        odets, omasks = oracle.detect(got[0][0], got[1][0], got[2][0], got[3][0], net.priors())
        assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets]
        assert np.array_equal(masks, omasks)
        # end to end against the oracle's own pipeline, over ALL detections
        acc = bench.accuracy_vs_oracle((dets, masks), (fdets, fmasks))
        assert acc["oracle_dets"] >= min_dets, (name, acc)
        assert acc["unmatched_oracle"] <= max_unmatched and acc["unmatched_engine"] <= max_unmatched, (name, acc)
        if name == "frc_balls" or acc["unmatched_oracle"] + acc["unmatched_engine"] == 0:
            assert acc["mask_iou_all"] >= 0.99, (name, acc)
        else:   # a full list: ties at the top-k cut may swap (bench.accuracy_vs_oracle: the *_above_cut pair leaves them out)
            assert acc["unmatched_above_cut"] == 0 and acc["mask_iou_above_cut"] >= 0.99, (name, acc)
        # and for EVERY oracle detection the engine's own softmax probability for that (class, prior) agrees
        conf = got[1][0]
        for d in fdets:
            z = conf[d["prior"]].astype(np.float64)
            pe = np.exp(z - z.max()); pe /= pe.sum()
            assert abs(pe[d["class_id"] + 1] - d["score"]) <= 5e-3, (name, d)


def test_batch_properties_at_full_size(eng550):
    eng, _ = eng550
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (8, S, S, 3), dtype=np.uint8)
    eng.set_input(frames)
    eng.evaluate()
    heads_a = [eng.output(i) for i in range(4)]
    dets_a = [eng.detections(f) for f in range(8)]
    # determinism: a second run (graph replay) is bitwise identical
    eng.set_input(frames)
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), heads_a[i])
    # frame-permutation equivariance: same n -> same tiles -> results move with their frame, bit for bit
    perm = np.array([3, 0, 7, 1, 6, 2, 5, 4])
    eng.set_input(frames[perm])
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), heads_a[i][perm])
    for f in range(8):
        d, m = eng.detections(f)
        assert d == dets_a[perm[f]][0] and np.array_equal(m, dets_a[perm[f]][1])
    # invariants of the tail
    for d, m in dets_a:
        assert 0 < len(d) <= 100
        s = [x["score"] for x in d]
        assert s == sorted(s, reverse=True) and min(s) > 0.05 and max(s) <= 1.0
        assert all(0 <= x["class_id"] < 80 and 0 <= x["prior"] < 19248 for x in d)
        assert np.isin(m, (0, 1)).all()
        for k, x in enumerate(d):      # every mask pixel lies inside its padded crop window
            x1, y1, x2, y2 = [v * 138 for v in x["box"]]
            ys, xs = np.nonzero(m[k])
            if len(xs):
                assert xs.min() >= min(x1, x2) - 1 - 1e-3 and xs.max() < max(x1, x2) + 1 + 1e-3
                assert ys.min() >= min(y1, y2) - 1 - 1e-3 and ys.max() < max(y1, y2) + 1 + 1e-3
        # Fast-NMS invariant on what survived: a kept box never overlaps an EARLIER kept box of its class > 0.5
        for j, a in enumerate(d):
            for b in d[:j]:
                if a["class_id"] != b["class_id"]:
                    continue
                iw = max(min(a["box"][2], b["box"][2]) - max(a["box"][0], b["box"][0]), 0)
                ih = max(min(a["box"][3], b["box"][3]) - max(a["box"][1], b["box"][1]), 0)
                ua = (a["box"][2] - a["box"][0]) * (a["box"][3] - a["box"][1]) + (b["box"][2] - b["box"][0]) * (b["box"][3] - b["box"][1]) - iw * ih
                assert ua <= 0 or iw * ih / ua <= 0.5 + 1e-6


def test_conv_linearity_exact_at_full_layer_size(built, oracle):
    """A full-size layer (the 138x138 256->256 3x3 protonet conv, 19.0 % of the FLOPs) through the
    op entry point: scaling the input by 2 scales the bias-free, activation-free output by exactly 2
    (power-of-two scaling commutes with every f32/f16 rounding), and two runs agree bitwise."""
    import yolact_amd as ya
    eng = ya.Engine(input_size=128, max_batch=1, use_graph=False)
    rng = np.random.default_rng(4)
    x = rng.normal(0, 1, (1, 138, 138, 256)).astype(np.float16).astype(np.float32)
    w = (rng.normal(0, 1, (256, 3, 3, 256)) / 48).astype(np.float16).astype(np.float32)
    b = np.zeros(256, np.float32)
    y1 = eng.op_conv2d(x, w, b, 1, 1, None, 0)
    y2 = eng.op_conv2d(2 * x, w, b, 1, 1, None, 0)
    normal = np.abs(y1) >= 2.0 ** -13            # f16 subnormals sit on an absolute grid: doubling is not exact there
    assert normal.mean() > 0.99 and np.array_equal((2 * y1)[normal], y2[normal])
    assert np.abs(2 * y1 - y2)[~normal].max(initial=0.0) <= 2.0 ** -23
    assert np.array_equal(eng.op_conv2d(x, w, b, 1, 1, None, 0), y1)
    # a sample of rows against the oracle (the whole layer would take the oracle a while)
    sub = oracle.conv2d(x[:, :10], w, b, 1, 1, None, 0, f16=True)
    assert np.abs(sub[:, :9] - y1[:, :9]).max() <= 2.0 ** -9 * max(1.0, np.abs(sub).max())
    eng.close()


def test_second_stream_changes_no_bit_over_many_steps(built):
    """The FPN's P4..P7 convolutions, the prediction head and the tail's K1-K3 run on a second stream beside the top-down
    chain and the protonet (tagged `side` ops, one fork per run, one join before the mask kernel). A missing
    dependency would show as a bit that differs from the one-stream step: 12 graph-replayed steps on changing frames at
    full size, batch 8 and batch 1, against an engine with tune.headfork_maxb = 0 and tune.tailfork = 0 - heads, prototypes,
    detections and masks bitwise equal every time."""
    import yolact_amd as ya
    rng = np.random.default_rng(21)
    for n in (8, 1):
        two = ya.Engine(input_size=S, max_batch=n, use_graph=True)
        one = ya.Engine(input_size=S, max_batch=n, use_graph=True, tune=dict(headfork_maxb=0, tailfork=0))
        blob = two.generate_weights(seed=1)
        two.load_weights(blob); one.load_weights(blob)
        for step in range(12):
            frames = rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8)
            outs = []
            for e in (two, one):
                e.set_input(frames)
                e.evaluate()
                outs.append(([e.output(i) for i in range(4)], [e.detections(f) for f in range(n)]))
            for a, b in zip(outs[0][0], outs[1][0]):
                assert np.array_equal(a, b), (n, step)
            for (da, ma), (db, mb) in zip(outs[0][1], outs[1][1]):
                assert da == db and np.array_equal(ma, mb), (n, step)
        two.close(); one.close()


def test_tail_exact_and_heads_repeatable_after_idle_gaps_on_the_two_stream_graph(eng550, oracle):
    """Round 5 found a compiled form of the candidate kernel (K1 with merged wide LDS reads AND the compiler's packed-f32
    exponentials) whose scores came out 1e-7 .. 1e-3 off - in lanes 48-63 of a wave only, only while proto3's launch ran beside
    it in the graph-replayed two-stream step at batch 1, a few detections in every second or third step once the device had idled
    (tools/study/tail_vs_oracle_repeat.py; detect.hip is built without packed-f32 arithmetic since). No other test in the suite has all three conditions except, by
    accident, the one-frame test above. This is the deliberate form: two frames alternated (so anything left over from the step
    before is WRONG data), 2 s of idle device before every step, and per step (a) the tail bit-exact on the engine's own head
    outputs, (b) heads and prototypes bit-equal to the same frame's earlier step."""
    import time
    eng, blob = eng550
    net = oracle.Net(50, S, 81, blob=blob)
    pri = net.priors()
    frames = [np.random.default_rng(5 + k).integers(0, 256, (1, S, S, 3), dtype=np.uint8) for k in range(2)]
    earlier = [None, None]
    for step in range(8):
        time.sleep(2.0)
        eng.set_input(frames[step & 1])
        eng.evaluate()
        got = [eng.output(i) for i in range(4)]
        dets, masks = eng.detections(0)
        assert len(dets) >= 90, (step, len(dets))
        odets, omasks = oracle.detect(got[0][0], got[1][0], got[2][0], got[3][0], pri)
        key = lambda ds: [(d["class_id"], d["prior"], d["score"], d["box"]) for d in ds]
        assert key(dets) == key(odets), step
        assert np.array_equal(masks, omasks), step
        if earlier[step & 1] is not None:
            for name, a, b in zip(("loc", "conf", "mask", "proto"), got, earlier[step & 1]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (step, name)
        earlier[step & 1] = got


# ---- configs[2]: batch 64, hipGraph steady state (BASELINE.json; tiles -> batch entries, src/yolact.rs:216-217) ----
LAYERS_550 = (("pool", 2e-3), ("c2", 6e-3), ("c3", 1e-2), ("c4", 1.5e-2), ("c5", 2e-2), ("lat5", 2e-2), ("lat4", 2e-2), ("lat3", 2e-2),
              ("p3", 2e-2), ("p4", 2e-2), ("p5", 2e-2), ("p6", 2.5e-2), ("p7", 2.5e-2), ("proto0", 2.5e-2), ("proto2", 2.5e-2),
              ("head_t0", 3e-2), ("head_t2", 3e-2), ("head_t4", 3e-2))


@pytest.fixture(scope="module")
def eng64(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=S, max_batch=64, use_graph=True)
    blob = e.generate_weights(seed=1)
    e.load_weights(blob)
    frames = np.random.default_rng(64).integers(0, 256, (64, S, S, 3), dtype=np.uint8)
    yield e, blob, frames
    e.close()


def test_batch64_launch_plans_are_the_ones_the_bench_runs(eng64):
    """The headline configuration picks its tiles and launch plans from M = 64 * P * Q: the streaming 1x1 tiles,
    the two-phase /rounds + /tail plan and the 256 + 128 channel split must really be launched at this size."""
    eng, _, frames = eng64
    eng.set_input(frames)
    eng.evaluate()
    names = [p["name"] for p in eng.profile(with_tail=True, reps=1)]
    assert eng.tuning()["plan_cus"] == 256                                   # MI355X: the device's own CU count
    assert sum("conv_igemm_f16<128,128,2,2,0,1>" in n for n in names) >= 20    # K1 streaming tiles (>= 1024 tiles each)
    assert any(n.endswith("/rounds") for n in names) and any(n.endswith("/tail") for n in names)
    assert any(n.endswith("/ch0-255") for n in names) and any(n.startswith("conv_igemm_f16<96,128,2,2,0,1,mfma16>[ml]") and n.endswith("/ch256-351") for n in names)   # (round 5: 96-channel tiles for the 95-channel remainder)
    # the largest conv runs with the 1x1 prototype conv in its epilogue: one launch for both, proto3 itself is never written
    assert any(n == "conv_igemm_f16<256,256,2,4,0,2,mfma16>[+1x1]:proto3+proto" for n in names) and not any(n.endswith(":proto") for n in names)
    import yolact_amd as ya
    with pytest.raises(ya.YhError):
        eng.tensor_frame("proto3", 0)
    assert not any("splitk" in n for n in names if ":p7" not in n)            # split-K is a small-batch plan


def test_batch64_determinism_permutation_and_tail_bit_exact(eng64, oracle):
    eng, blob, frames = eng64
    eng.set_input(frames)
    eng.evaluate()
    heads_a = [eng.output(i) for i in range(4)]
    dets_a = [eng.detections(f) for f in range(64)]
    # determinism under graph replay
    eng.set_input(frames)
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), heads_a[i])
    # frame-permutation equivariance over the 64 slots: a row's bits do not depend on the tile, round or launch
    # phase (rounds / tail, channel split) that computed it
    perm = np.random.default_rng(1).permutation(64)
    eng.set_input(frames[perm])
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), heads_a[i][perm])
    for f in range(64):
        d, m = eng.detections(f)
        assert d == dets_a[perm[f]][0] and np.array_equal(m, dets_a[perm[f]][1])
    # the detection tail of EVERY frame, bit for bit, against the oracle on the engine's own head outputs
    pri = oracle.Net(50, S, 81, blob=blob).priors()
    total = 0
    for f in range(64):
        dets, masks = dets_a[f]
        odets, omasks = oracle.detect(heads_a[0][f], heads_a[1][f], heads_a[2][f], heads_a[3][f], pri)
        assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets], f
        assert np.array_equal(masks, omasks), f
        total += len(dets)
    assert total >= 64 * 50


def test_batch64_one_slot_layer_by_layer_vs_oracle(eng64, oracle):
    """One frame at a random slot of the batch-64 run against the oracle's forward: every named layer the production
    path materialises (the layer-by-layer check that used to exist only at 128 x 128), the heads, and the detections."""
    import bench
    eng, blob, frames = eng64
    slot = 37
    eng.set_input(frames)
    eng.evaluate()
    net = oracle.Net(50, S, 81, blob=blob)
    want = net.forward(frames[slot:slot + 1], f16=True)
    for name, tol in LAYERS_550:
        a, b = eng.tensor_frame(name, slot), net.get(name)[0]
        assert a.shape == b.shape, name
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
    got = [eng.output(i)[slot] for i in range(4)]
    for tn, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        assert np.abs(a - b[0]).max() <= 0.03 * max(1.0, np.abs(b).max()), tn
        assert np.sqrt(((a - b[0]) ** 2).mean()) <= 5e-3 * np.sqrt((b ** 2).mean()) + 1e-4, tn
    fd = oracle.detect(want[0][0], want[1][0], want[2][0], want[3][0], net.priors())
    # A uniform-noise frame under seeded weights fills the list (100 detections) with scores that differ in the 4th decimal
    # at the cut, so which candidates take the last places is decided by f16 summation order (this slot: ranks 98 / 99 swap,
    # and the engine-only one has a frame-sized box) - the acceptance frame (test_one_frame_550_vs_oracle) has no such tie.
    # Asserted: everything that is not within 2 % of the cut score is matched and its masks agree; at most 3 swaps at the cut.
    acc = bench.accuracy_vs_oracle(eng.detections(slot), fd)
    assert acc["oracle_dets"] >= 50 and acc["unmatched_oracle"] <= 3 and acc["unmatched_engine"] <= 3, acc
    assert acc["unmatched_above_cut"] == 0 and acc["dets_above_cut"] >= 80 and acc["mask_iou_above_cut"] >= 0.99, acc


# ---- configs[4] geometry in f16: YOLACT-700 ResNet-101, one frame ----
def test_yolact700_r101_one_frame_vs_oracle(built, oracle):
    import bench
    import yolact_amd as ya
    S7 = 700
    eng = ya.Engine(input_size=S7, backbone=101, max_batch=1, use_graph=True)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    assert eng.P == 30963 and (eng.hp, eng.wp) == (176, 176)                  # levels 88/44/22/11/6
    assert abs(eng.flops_per_frame() / 1e9 - 262.93) < 0.1
    net = oracle.Net(101, S7, 81, blob=blob)
    img = np.random.default_rng(7).integers(0, 256, (1, S7, S7, 3), dtype=np.uint8)
    got, dets, want, fd = _vs_oracle(eng, net, oracle, img)
    for name, tol in (("c2", 6e-3), ("c3", 1e-2), ("c4", 2.5e-2), ("c5", 3e-2), ("p3", 3e-2), ("p7", 3.5e-2), ("proto3", 4e-2), ("head_t0", 4e-2)):
        a, b = eng.tensor_frame(name, 0), net.get(name)[0]
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
    for tn, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        assert np.abs(a - b).max() <= 0.04 * max(1.0, np.abs(b).max()), tn
        assert np.sqrt(((a - b) ** 2).mean()) <= 8e-3 * np.sqrt((b ** 2).mean()) + 1e-4, tn
    odets, omasks = oracle.detect(got[0][0], got[1][0], got[2][0], got[3][0], net.priors())
    assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets[0]] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets]
    assert np.array_equal(dets[1], omasks)
    acc = bench.accuracy_vs_oracle(dets, fd)
    assert acc["unmatched_oracle"] <= 3 and acc["unmatched_engine"] <= 3, acc
    if acc["unmatched_oracle"] + acc["unmatched_engine"] == 0:
        assert acc["mask_iou_all"] is None or acc["mask_iou_all"] >= 0.99, acc
    else:
        assert acc["unmatched_above_cut"] == 0 and (acc["mask_iou_above_cut"] is None or acc["mask_iou_above_cut"] >= 0.99), acc
    eng.close()


# ---- configs[4] itself: YOLACT-700 ResNet-101 with fp8 operands (yh_config.precision = YH_PRECISION_FP8) ----
def test_yolact700_r101_fp8_one_frame_vs_fp8_oracle(built, oracle):
    """The fp8 forward at configs[4]'s geometry against the oracle's fp8 mode with the engine's calibrated scales (bounds
    and their reasons: tests/test_gpu_fp8.py), the tail bit-exact on the engine's heads, and the
    frame-slot independence of a batch of 3 (at this batch size most of the 36 E4M3 launches take the 64 x 64 fp8 tile, the 88 x 88
    and 176 x 176 layers the 128 x 128 one; every tile, layer by layer on identical inputs: tests/test_gpu_fp8_sweep.py)."""
    import yolact_amd as ya
    S7 = 700
    eng = ya.Engine(input_size=S7, backbone=101, max_batch=3, use_graph=True, precision=ya.PRECISION_FP8)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (3, S7, S7, 3), dtype=np.uint8)
    eng.set_input(frames)
    eng.fp8_calibrate()
    layers = eng.fp8_layers()
    assert len(layers) == 36 and {"l3b22_b", "l4b2_b", "p7", "proto3", "head_t"} <= {n for n, _ in layers}
    eng.evaluate()
    names = [p["name"] for p in eng.profile(with_tail=False, reps=1)]
    assert sum(n.startswith("conv_igemm_fp8<") for n in names) == 36
    heads = [eng.output(i) for i in range(4)]
    dets = [eng.detections(f) for f in range(3)]
    net = oracle.Net(101, S7, 81, blob=blob)
    lay = {}
    for name, sc in eng.fp8_channel_scales():             # one scale per input channel (round 4), folded into the weights on both sides
        for nm in ([f"{name}{l}" for l in range(5)] if name == "head_t" else [name]):
            lay[nm] = sc
    net.set_fp8(lay)
    want = net.forward(frames[1:2], f16=True)
    for tn, a, b in zip(("loc", "conf", "mask", "proto"), heads, want):
        mx = float(np.abs(a[1] - b[0]).max() / max(1.0, np.abs(b).max()))
        rms = float(np.sqrt(((a[1] - b[0]) ** 2).mean()) / np.sqrt((b ** 2).mean()))
        # deeper than the 160-pixel R50 case (36 chained E4M3 layers): rms <= 12 % of rms for every head; max <= 20 % of absmax,
        # except the tanh mask coefficients, where a logit near 0 with a flipped sign is a legitimate O(1) outlier among 10^6
        # values (measured: max 1.2, 99.9th percentile 0.54): there the 99th percentile is bounded instead
        q99 = float(np.quantile(np.abs(a[1] - b[0]), 0.99))
        assert rms <= 0.12 and (q99 <= 0.35 if tn == "mask" else mx <= 0.2), (tn, mx, rms, q99)
    pri = net.priors()
    for f in range(3):
        odets, omasks = oracle.detect(heads[0][f], heads[1][f], heads[2][f], heads[3][f], pri)
        assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets[f][0]] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets]
        assert np.array_equal(dets[f][1], omasks)
    perm = np.array([2, 0, 1])
    eng.set_input(frames[perm])
    eng.evaluate()
    for i in range(4):
        assert np.array_equal(eng.output(i), heads[i][perm])
    eng.close()
