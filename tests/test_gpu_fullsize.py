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


def test_one_frame_550_vs_oracle(eng550, oracle, golden_dir):
    """Restated acceptance target at full size: frc_balls.png resized to 550x550,
    HIP engine vs CPU oracle in f16-storage mode. NOT parity with CPU tflite (model file absent)."""
    from PIL import Image
    eng, blob = eng550
    img = np.asarray(Image.open(os.path.join(golden_dir, "frc_balls.png")).convert("RGB").resize((S, S), Image.BILINEAR))[None]
    eng.set_input(img)
    eng.evaluate()
    got = [eng.output(i) for i in range(4)]
    dets, masks = eng.detections(0)
    net = oracle.Net(50, S, 81, blob=blob)
    want = net.forward(img, f16=True)
    for name, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        assert np.abs(a - b).max() <= 0.03 * max(1.0, np.abs(b).max()), name
        assert np.sqrt(((a - b) ** 2).mean()) <= 5e-3 * np.sqrt((b ** 2).mean()) + 1e-4, name
    # tail bit-exact on the engine's own head outputs
    odets, omasks = oracle.detect(got[0][0], got[1][0], got[2][0], got[3][0], net.priors())
    assert [(d["class_id"], d["prior"], d["score"], d["box"]) for d in dets] == [(d["class_id"], d["prior"], d["score"], d["box"]) for d in odets]
    assert np.array_equal(masks, omasks)
    # end to end against the oracle's own pipeline: matched detections, class ids equal, mask IoU >= 0.99
    fdets, fmasks = oracle.detect(want[0][0], want[1][0], want[2][0], want[3][0], net.priors())
    key = {(d["class_id"], d["prior"]): i for i, d in enumerate(dets)}
    matched = [(key[(d["class_id"], d["prior"])], j) for j, d in enumerate(fdets) if (d["class_id"], d["prior"]) in key]
    # Which of ~19 000 near-tied candidates survive Fast-NMS and the top-100 cut is a discrete decision:
    # with seeded (untrained) weights many same-class neighbours sit at IoU ~ 0.5 with scores that
    # differ in the 4th digit, so a few survivors flip on summation-order noise (8-11 of 100 here,
    # whatever the tile choice). What must hold: most detections match, and for EVERY oracle detection
    # the engine's own softmax probability for that (class, prior) agrees with the oracle's score.
    assert len(fdets) > 0 and len(matched) >= 0.8 * len(fdets)
    conf = got[1][0]
    for d in fdets:
        z = conf[d["prior"]].astype(np.float64)
        pe = np.exp(z - z.max()); pe /= pe.sum()
        assert abs(pe[d["class_id"] + 1] - d["score"]) <= 5e-3, d
    inter = sum(int((masks[i] & fmasks[j]).sum()) for i, j in matched)
    union = sum(int((masks[i] | fmasks[j]).sum()) for i, j in matched)
    assert inter / max(union, 1) >= 0.99


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
