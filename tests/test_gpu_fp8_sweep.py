"""-m gpu: configs[4] layer by layer (BASELINE.json: YOLACT-700 ResNet-101, fp8 operands on the fp8 MFMA).

Teacher-forced sweep of ALL 36 E4M3 launches at full geometry: each layer's input is the ENGINE's own tensor of that run
(quantised by the same function its producer's epilogue applies), the oracle convolves the decoded operands in f32, and the
engine's output of that one layer must agree within 2 f16 ulp + 2^-10 of the sum of |products| - the summation-order bound
of tests/test_gpu_fp8.py, where only three layer kinds at 160 pixels were covered. Chained end to end, two fp8
implementations drift apart by E4M3 code flips; teacher forcing removes exactly that and leaves the kernel.
The reference's model is quantised end to end too (uint8: data/FRC_model_edgetpu.log:7-19); nothing in the reference pins
E4M3 arithmetic: parity unpinned, engine == oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S7, NB = 700, 5   # batch 5: p3 / proto0-2 take the 128 x 128 fp8 tile, proto3 and the multi-level head trunk the 256 x 256 one, the rest 64 x 64


def conv_indices(backbone):
    """Layer name -> index in the canonical weight blob ("Weight blob")."""
    idx, ci = {}, 1
    for L, nb in enumerate((3, 4, 23, 3) if backbone == 101 else (3, 4, 6, 3)):
        for b in range(nb):
            idx[f"l{L + 1}b{b}_a"], idx[f"l{L + 1}b{b}_b"], idx[f"l{L + 1}b{b}_c"] = ci, ci + 1, ci + 2
            ci += 4 if b == 0 else 3
    for k, name in enumerate(("lat5", "lat4", "lat3", "p5", "p4", "p3", "p6", "p7", "proto0", "proto1", "proto2", "proto3", "proto", "head_t")):
        idx[name] = ci + k
    return idx


def fp8_layer_table(backbone):
    """(engine layer name, input tensor, output tensor, stride, relu) of every E4M3 launch but the multi-level head trunk."""
    t = []
    for L, nb in ((3, 23 if backbone == 101 else 6), (4, 3)):
        for b in range(nb):
            t.append((f"l{L}b{b}_b", f"l{L}b{b}_a", f"l{L}b{b}_b", 2 if b == 0 else 1, True))
    t += [("p5", "lat5", "p5", 1, True), ("p6", "p5", "p6", 2, False), ("p7", "p6", "p7", 2, False), ("p4", "lat4", "p4", 1, True), ("p3", "lat3", "p3", 1, True),
          ("proto0", "p3", "proto0", 1, True), ("proto1", "proto0", "proto1", 1, True), ("proto2", "proto1", "proto2", 1, True), ("proto3", "proto_up", "proto3", 1, True)]
    return t


class Forced:
    """Teacher-forced expectation of one fp8 convolution from the engine's own input tensor."""

    def __init__(self, oracle, blob, backbone):
        import bench
        self.O, self.convs, self.idx, self.table = oracle, bench.parse_blob(blob), conv_indices(backbone), oracle.e4m3_decode_table()
        self._wq = {}

    def weights(self, name, s_c):
        """The layer's weights with its input tensor's channel scales folded in along K (t = w * s_c, one f32 multiplication), then
        one scale per output channel and E4M3 codes of t / s_w - engine.hip: refresh_fp8_scales, oracle/orc_net.c: run_conv."""
        if name not in self._wq:
            w, b = self.convs[self.idx[name]]
            t = (w.astype(np.float32) * s_c.astype(np.float32)).astype(np.float32)
            aw = np.abs(t).reshape(t.shape[0], -1).max(1).astype(np.float32)
            sw = np.where(aw > 0, aw / np.float32(448.0), np.float32(1.0)).astype(np.float32)
            inv = (np.float32(1.0) / sw).astype(np.float32)
            wq = np.stack([self.table[self.O.quantize_e4m3(t[o], float(inv[o]))] for o in range(t.shape[0])]).astype(np.float32)
            self._wq[name] = (wq, sw, b)
        return self._wq[name]

    def expect(self, name, x_f32, s_c, stride, relu):
        """x_f32: the engine's input tensor [1][h][w][c] (f16 values, or exactly decoded E4M3 values); s_c: its channel scales [c].
        Returns (want f16-rounded, sum of |products| in output units)."""
        s_c = np.asarray(s_c, np.float32)
        inv = (np.float32(1.0) / s_c).astype(np.float32)         # the epilogue multiplies by 1 / s[c] (engine.hip: y8_inv)
        xq = self.table[self.O.quantize_e4m3((x_f32.astype(np.float32) * inv).astype(np.float32), 1.0)].astype(np.float32)
        wq, sw, b = self.weights(name, s_c)
        zero = np.zeros(wq.shape[0], np.float32)
        acc = self.O.conv2d(xq, wq, zero, stride, 1, None, 0, f16=False)
        mag = self.O.conv2d(np.abs(xq), np.abs(wq), zero, stride, 1, None, 0, f16=False) * sw
        v = acc * sw + b
        if relu:
            v = np.maximum(v, 0)
        return v.astype(np.float16).astype(np.float32), mag


def _check(layer, got, want, mag, report):
    ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
    err = np.abs(got - want)
    assert got.shape == want.shape, (layer, got.shape, want.shape)
    ratio = float((err / np.maximum(mag, 1e-30)).max())
    report.append(f"{layer}: max |err| {err.max():.5f}, {100 * (err > 2 * ulp).mean():.3f} % beyond 2 ulp, max err / sum|products| = 2^{np.log2(max(ratio, 1e-30)):.1f}")
    bad = err > 2 * ulp + 2.0 ** -10 * mag + 1e-6
    assert not bad.any(), (layer, int(bad.sum()), float(err.max()), ratio)


@pytest.fixture(scope="module")
def r101_fp8(built, oracle):
    """The production engine P (every fusion on, E4M3-only tensors) and the same configuration with debug_tensors = 1 (D: every
    tensor also in f16), both at batch 5 with P's calibrated scales."""
    import yolact_amd as ya
    P = ya.Engine(input_size=S7, backbone=101, max_batch=NB, use_graph=True, precision=ya.PRECISION_FP8)
    blob = P.generate_weights(seed=1)
    P.load_weights(blob)
    frames = np.random.default_rng(17).integers(0, 256, (NB, S7, S7, 3), dtype=np.uint8)
    import bench
    balls = bench.acceptance_frame(S7)              # the reference's own test image (tests/golden/frc_balls.png) as the last frame
    if balls is not None:
        frames[NB - 1] = balls[0]
    P.set_input(frames)
    P.fp8_calibrate()
    P.evaluate()
    D = ya.Engine(input_size=S7, backbone=101, max_batch=NB, use_graph=False, precision=ya.PRECISION_FP8, debug_tensors=True)
    D.load_weights(blob)
    assert [n for n, _ in D.fp8_layers()] == [n for n, _ in P.fp8_layers()]
    for i, (_, sc) in enumerate(P.fp8_channel_scales()):
        D.fp8_set_layer_scale(i, sc)
    D.set_input(frames)
    D.evaluate()
    yield P, D, blob, frames, Forced(oracle, blob, 101)
    P.close(); D.close()


def test_all_36_fp8_launches_teacher_forced_r101_700(r101_fp8):
    P, D, blob, frames, forced = r101_fp8
    sc = dict(D.fp8_channel_scales())
    assert len(sc) == 36 and all(np.array_equal(v, dict(P.fp8_channel_scales())[n]) for n, v in sc.items())
    # which tile each launch took (yh_profile_run names): all three fp8 tiles and the multi-level 256 x 256 form are in the sweep
    names = [p["name"] for p in D.profile(with_tail=False, reps=1)]
    fp8_names = [n for n in names if n.startswith("conv_igemm_fp8<")]
    assert len(fp8_names) == 36, fp8_names
    tile_of = {n.split(":")[1].split("/")[0].split("+")[0]: n.split(":")[0] for n in fp8_names}
    assert tile_of["proto3"] == "conv_igemm_fp8<256,256,2,4>" and tile_of["head_t"] == "conv_igemm_fp8<256,256,2,4>[ml]", tile_of
    assert tile_of["p3"] == tile_of["proto1"] == "conv_igemm_fp8<128,128,2,2>", tile_of
    assert tile_of["l3b7_b"] == tile_of["l4b1_b"] == tile_of["p6"] == "conv_igemm_fp8<64,64,2,2>", tile_of
    report = []
    for f in (0, NB - 1):                       # first and last frame of the batch: every tile row range is touched at both ends
        for layer, xin, yout, stride, relu in fp8_layer_table(101):
            if f and layer.startswith("l3b") and layer not in ("l3b0_b", "l3b11_b", "l3b22_b"):
                continue                        # (the 23 identical layer-3 shapes: all on frame 0, three on the last frame)
            want, mag = forced.expect(layer, D.tensor_frame(xin, f)[None], sc[layer], stride, relu)
            _check(f"{layer}[frame {f}, {tile_of[layer]}]", D.tensor_frame(yout, f)[None], want, mag, report)
        for l in range(5):                      # the shared head trunk: one multi-level launch, checked level by level
            want, mag = forced.expect("head_t", D.tensor_frame(f"p{l + 3}", f)[None], sc["head_t"], 1, True)
            _check(f"head_t{l}[frame {f}, {tile_of['head_t']}]", D.tensor_frame(f"head_t{l}", f)[None], want, mag, report)
    print("\n".join(report))
    # the multi-level 128 x 128 fp8 form (head trunk at batches below ~4): three frames on the same handles
    D.set_input(frames[:3]); D.evaluate()
    names3 = [p["name"] for p in D.profile(with_tail=False, reps=1)]
    assert any(n.startswith("conv_igemm_fp8<128,128,2,2>[ml]:head_t") for n in names3), [n for n in names3 if "head_t" in n]
    rep3 = []
    for l in (0, 2, 4):
        want, mag = forced.expect("head_t", D.tensor_frame(f"p{l + 3}", 2)[None], sc["head_t"], 1, True)
        _check(f"head_t{l}[batch 3, frame 2, 128x128 ml]", D.tensor_frame(f"head_t{l}", 2)[None], want, mag, rep3)
    print("\n".join(rep3))
    D.set_input(frames); D.evaluate()


def test_production_fp8_engine_e4m3_only_tensors_and_fused_prototype_conv(r101_fp8, oracle):
    """The production plan (no debug tensors): tensors that only fp8 convolutions read exist ONLY as E4M3 codes, and the
    largest conv runs on the 256 x 256 fp8 tile with the 1x1 prototype conv in its epilogue (proto3 is never written).
    * E4M3-only outputs (p3..p7, proto0, proto1): the decoded engine tensor against the teacher-forced f16 expectation, which
      the producer's epilogue then quantises - |difference| <= the summation bound + half an E4M3 step of the value (2^-4
      relative; 2^-10 of the scale below the normal range) - and every decoded value is a code value times the scale.
    * proto3 + proto fused: the prototypes against conv1x1(relu(teacher-forced proto3)) with the first conv's bound pushed
      through the second conv's |weights|."""
    import bench
    import yolact_amd as ya
    P, D, blob, frames, forced = r101_fp8
    sc = dict(P.fp8_channel_scales())
    names = [p["name"] for p in P.profile(with_tail=False, reps=1)]
    assert "conv_igemm_fp8<256,256,2,4>[+1x1]:proto3+proto" in names, [n for n in names if "proto" in n]
    with pytest.raises(ya.YhError):
        P.tensor_frame("proto3", 0)
    consumer = {"p3": "proto0", "p4": "head_t", "p5": "p6", "p6": "p7", "p7": "head_t", "proto0": "proto1", "proto1": "proto2"}
    codes = np.sort(forced.table[np.isfinite(forced.table)].astype(np.float64))
    report = []
    f = 1
    for layer, xin, yout, stride, relu in fp8_layer_table(101):
        if yout not in consumer:
            continue
        want, mag = forced.expect(layer, P.tensor_frame(xin, f)[None], sc[layer], stride, relu)
        got = P.tensor_frame(yout, f)[None]
        s_out = sc[consumer[yout]].astype(np.float32)              # the channel scales of the tensor this layer writes
        q = (got.astype(np.float64) / s_out).ravel()               # every decoded value is a code value times its channel's scale
        near = np.abs(codes[np.clip(np.searchsorted(codes, q), 1, len(codes) - 1)] - q)
        near = np.minimum(near, np.abs(codes[np.clip(np.searchsorted(codes, q), 1, len(codes) - 1) - 1] - q))
        assert (near <= 1e-5 * np.maximum(1.0, np.abs(q))).all(), layer
        ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
        half_step = np.maximum(np.abs(want) * 2.0 ** -4, s_out * 2.0 ** -10)
        err = np.abs(got - np.clip(want, -448 * s_out, 448 * s_out))
        report.append(f"{layer} (E4M3-only output): max |err| {err.max():.5f} vs half step {half_step.max():.5f}")
        assert (err <= 2 * ulp + 2.0 ** -10 * mag + half_step * 1.001 + 1e-6).all(), (layer, float(err.max()))
    # the fused pair
    want3, mag3 = forced.expect("proto3", P.tensor_frame("proto_up", f)[None], sc["proto3"], 1, True)
    w2, b2 = bench.parse_blob(blob)[forced.idx["proto"]]
    want = np.maximum(oracle.conv2d(want3, w2, b2, 1, 0, None, 0, f16=False), 0).astype(np.float16).astype(np.float32)
    tol3 = 2 * np.maximum(np.abs(want3), 2.0 ** -14) * 2.0 ** -10 + 2.0 ** -10 * mag3
    push = oracle.conv2d(tol3, np.abs(w2), np.zeros_like(b2), 1, 0, None, 0, f16=False)
    got = P.output(3)[f][None]
    err = np.abs(got - want)
    report.append(f"proto3+proto fused: max |err| {err.max():.5f}, bound max {float((push + 2 * np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10).max()):.5f}")
    assert (err <= push + 2 * np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10 + 1e-6).all(), float(err.max())
    print("\n".join(report))


def test_fp8_detections_vs_fp8_oracle_above_the_head_noise(r101_fp8, oracle):
    """Detection level, end to end (no teacher forcing): the fp8 engine against the oracle's fp8 forward mode with the engine's
    scales, on a noise frame and on the reference's test image at 700 x 700. Chained over 36 E4M3 layers the two drift apart by
    code flips: measured, the conf logits differ by 3-4 % rms and a detection's probability by up to e^0.37.
    Asserted, with that noise as the margin:
      * continuity - for EVERY detection of either side, the other side's own softmax probability of that (class, prior) is
        within a factor e^0.6 of its score;
      * decisions - every detection whose score clears the 0.05 threshold by that factor (score > 0.091) is a detection of the
        other side too, up to max(3, 20 %) of them (measured: 0 of 16 and 3 of 19 on the noise frame - their probabilities agree
        (continuity), but Fast-NMS is a second discrete decision: with untrained box regressions a neighbour's IoU straddles 0.5);
      * masks - over matched pairs with at least 500 mask pixels: pixel-weighted IoU and median IoU >= 0.5 on the noise frame,
        >= 0.65 / 0.6 on frc_balls (the prototypes differ by 6 % rms around the sigmoid's 0.5 level).
    Against the F16 oracle - the price of the precision, not a parity claim - the figures are printed for the noise frame and,
    since round 4, asserted on frc_balls (below)."""
    P, D, blob, frames, forced = r101_fp8
    P.set_input(frames); P.evaluate()
    heads = [P.output(i) for i in range(4)]
    net = oracle.Net(101, S7, 81, blob=blob)
    lay = {}
    for name, sc in P.fp8_channel_scales():
        for nm in ([f"{name}{l}" for l in range(5)] if name == "head_t" else [name]):
            lay[nm] = sc
    pri = net.priors()
    FACTOR = float(np.exp(0.6))

    def prob(conf_row, cls):
        z = conf_row.astype(np.float64)
        pe = np.exp(z - z.max())
        return float(pe[cls + 1] / pe.sum())
    for f in (0, NB - 1):
        net.set_fp8(lay)
        w8 = net.forward(frames[f:f + 1], f16=True)
        net.set_fp8(None)
        od, om = oracle.detect(w8[0][0], w8[1][0], w8[2][0], w8[3][0], pri)
        ed, em = P.detections(f)
        assert len(od) >= 5 and len(ed) >= 5, (f, len(od), len(ed))
        assert len(od) < 100 and len(ed) < 100        # (the list is not full: the cut is the score threshold itself)
        ek = {(d["class_id"], d["prior"]): i for i, d in enumerate(ed)}
        ok_ = {(d["class_id"], d["prior"]): j for j, d in enumerate(od)}
        worst = 0.0
        for d in od:
            worst = max(worst, abs(np.log(prob(heads[1][f][d["prior"]], d["class_id"]) / d["score"])))
        for d in ed:
            worst = max(worst, abs(np.log(prob(w8[1][0][d["prior"]], d["class_id"]) / d["score"])))
        strong_o = [d for d in od if d["score"] > 0.05 * FACTOR]
        strong_e = [d for d in ed if d["score"] > 0.05 * FACTOR]
        miss_o = sum((d["class_id"], d["prior"]) not in ek for d in strong_o)
        miss_e = sum((d["class_id"], d["prior"]) not in ok_ for d in strong_e)
        ious = []
        for key, i in ek.items():
            j = ok_.get(key)
            if j is None:
                continue
            x, y = em[i] > 0, om[j] > 0
            u = int((x | y).sum())
            if u >= 500:
                ious.append((int((x & y).sum()) / u, u))
        v, u = np.array([t[0] for t in ious]), np.array([t[1] for t in ious])
        weighted = float((v * u).sum() / u.sum())
        print(f"frame {f}: oracle {len(od)} / engine {len(ed)} detections, matched {len(set(ek) & set(ok_))}; max |log prob ratio| {worst:.3f}; "
              f"above margin: oracle {len(strong_o)} (unmatched {miss_o}), engine {len(strong_e)} (unmatched {miss_e}); "
              f"{len(v)} mask pairs: pixel-weighted IoU {weighted:.3f}, median {np.median(v):.3f}")
        real = f == NB - 1                      # the reference's test image: a handful of confident detections, not a list of threshold cases
        # (round 4, per-input-channel scales - measured: noise 0.508, 1 / 14 and 2 / 15 unmatched, 0.60 / 0.65; frc_balls 0.234, 0 / 1 and 0 / 2, 0.76 / 0.74)
        assert worst <= (0.35 if real else 0.6), (f, worst)
        assert miss_o <= max(3, len(strong_o) // 5) and miss_e <= max(3, len(strong_e) // 5), (f, miss_o, len(strong_o), miss_e, len(strong_e))
        assert len(v) >= 3 and weighted >= (0.65 if real else 0.5) and np.median(v) >= (0.6 if real else 0.5), (f, len(v), weighted, float(np.median(v)))
        # the same engine detections against the F16 oracle - the price of the precision. Reported for the noise frame; ASSERTED
        # (round 4) on the reference's test image at 700 x 700: every detection of the f16 oracle but at most one is a detection of
        # the fp8 engine, with a mask IoU over the matched pairs of at least 0.70 (measured: 8 of 8, 0.78)
        import bench
        w16 = net.forward(frames[f:f + 1], f16=True)
        acc = bench.accuracy_vs_oracle((ed, em), oracle.detect(w16[0][0], w16[1][0], w16[2][0], w16[3][0], pri))
        print(f"frame {f}: fp8 engine vs f16 oracle: matched {acc['matched_class_and_prior']} of {acc['oracle_dets']} (engine {acc['engine_dets']}), mask IoU matched {acc['mask_iou_matched']}, all {acc['mask_iou_all']}")
        if real:
            assert acc["oracle_dets"] >= 5 and acc["unmatched_oracle"] <= 1 and acc["unmatched_engine"] <= 2 and acc["mask_iou_matched"] >= 0.70, acc


def _fp8_layer_map(eng):
    lay = {}
    for name, sc in eng.fp8_channel_scales():
        for nm in ([f"{name}{l}" for l in range(5)] if name == "head_t" else [name]):
            lay[nm] = sc
    return lay


def test_fp8_share_of_eight_frames_pooled_against_its_checker(built, oracle):
    """configs[4]'s own per-GPU share - YOLACT-700 R101, fp8 precision, EIGHT frames - at detection level, pooled the way the
    driver-timed bench line pools it (`configs4.accuracy.*_all_frames`: bench.fp8_vs_oracles / pool_accuracy): the fp8 engine
    against the oracle's fp8 forward mode with the engine's own calibrated scales, over all eight frames (seven noise frames and the
    reference's test image). Round 4's test looked at two frames; one noise frame is a handful of decisions near the 0.05
    threshold. Floors just under what is measured - on these eight frames 171 matched of the checker's 207 (0.83; per frame 21 of 26 ...
    27 of 30), matched-pair mask IoU 0.75, 229 engine detections (round 5's calibration: twice the channel maximum, at least a sixteenth
    of the tensor's; with round 4's bare channel maxima: 179 of 232 = 0.77, 0.70; with the floor at an eighth: 168 of 219 = 0.77, 0.74 -
    the spread of one scheme change IS the noise of these figures); the driver's line of round 4, on the bench's own frames: 188 of 235
    (0.80), 0.66 -:
      matched fraction of the checker's detections >= 0.74, detection-weighted mean of the matched-pair mask IoU >= 0.62,
      the engine's own detection count within 15 % of the checker's (spurious detections are decisions too)."""
    import bench
    import yolact_amd as ya
    n = 8
    eng = ya.Engine(input_size=S7, backbone=101, max_batch=n, use_graph=True, precision=ya.PRECISION_FP8)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    frames = np.random.default_rng(8).integers(0, 256, (n, S7, S7, 3), dtype=np.uint8)
    balls = bench.acceptance_frame(S7)
    if balls is not None:
        frames[n - 1] = balls[0]
    eng.set_input(frames)
    eng.fp8_calibrate()
    eng.evaluate()
    names = [p["name"] for p in eng.profile(with_tail=True, reps=1)]
    assert sum(nm.startswith("bneck_xn_f16:") for nm in names) == 21            # the share's own launch plan (242 tiles per layer-3 launch)
    net = oracle.Net(101, S7, 81, blob=blob)
    pri = net.priors()
    net.set_fp8(_fp8_layer_map(eng))
    per = []
    for f in range(n):
        h8 = net.forward(frames[f:f + 1], f16=True)
        per.append(bench.accuracy_vs_oracle(eng.detections(f), oracle.detect(h8[0][0], h8[1][0], h8[2][0], h8[3][0], pri)))
    net.set_fp8(None)
    pooled = bench.pool_accuracy(per)
    print("pooled over 8 frames, fp8 engine vs fp8-mode checker:", {k: v for k, v in pooled.items() if not k.endswith("_per_frame")})
    print("per frame matched / checker / engine:", [(r["matched_class_and_prior"], r["oracle_dets"], r["engine_dets"]) for r in per])
    assert pooled["oracle_dets"] >= 100 and pooled["matched_fraction_of_oracle"] >= 0.74, pooled
    assert pooled["mask_iou_matched_mean"] >= 0.62, pooled
    assert abs(pooled["engine_dets"] - pooled["oracle_dets"]) <= 0.15 * pooled["oracle_dets"], pooled
    eng.close()


def test_fp8_channel_scales_calibrated_on_other_frames(built, oracle):
    """the per-input-channel activation scales (the default since round 4) are maxima of the calibration frames with no
    headroom - a channel that is quiet there and active later saturates at 448 s[c], a risk one scale per tensor does not have - and
    every accuracy figure of the suite calibrated and evaluated on the same frames. Held-out check, both directions: calibrate on
    noise frames and evaluate on the reference's test image at 700 x 700, calibrate on that image and evaluate on a noise frame;
    against the F16 oracle, the per-channel scheme must do no worse than one scale per tensor calibrated the same way (matched
    detections, up to two - one noise frame's list is some thirty decisions near the threshold; matched-pair mask IoU, up to 0.05).
    Measured with round 4's scales (bare channel maxima): calibrated on the test image, evaluated on noise 21 of 31 matched, IoU 0.63,
    against 25 and 0.74 for one scale per tensor - the risk was real. With round 5's headroom (twice the channel maximum, at least a
    sixteenth of the tensor's): 23 of 31, 0.77; the other direction 8 of 8, 0.76 (per tensor: 8 of 8, 0.80).
    Nothing in the reference pins E4M3: parity unpinned."""
    import bench
    import yolact_amd as ya
    balls = bench.acceptance_frame(S7)
    if balls is None:
        pytest.skip("tests/golden/frc_balls.png is missing")
    noise = np.random.default_rng(23).integers(0, 256, (3, S7, S7, 3), dtype=np.uint8)
    net = oracle.Net(101, S7, 81, seed=1)
    pri = net.priors()

    def f16_dets(fr):
        h = net.forward(fr, f16=True)
        return oracle.detect(h[0][0], h[1][0], h[2][0], h[3][0], pri)
    want = {"balls": f16_dets(balls[:1]), "noise": f16_dets(noise[:1])}
    res = {}
    for per_tensor in (False, True):
        eng = ya.Engine(input_size=S7, backbone=101, max_batch=3, use_graph=True, precision=ya.PRECISION_FP8, fp8_per_tensor=per_tensor)
        eng.load_weights(eng.generate_weights(seed=1))
        for calib, ev, key in ((noise, balls[:1], "balls"), (np.repeat(balls[:1], 3, 0), noise[:1], "noise")):
            eng.set_input(calib)
            eng.fp8_calibrate()
            eng.set_input(ev)
            eng.evaluate()
            res[(per_tensor, key)] = bench.accuracy_vs_oracle(eng.detections(0), want[key])
        eng.close()
    for key in ("balls", "noise"):
        ch, pt = res[(False, key)], res[(True, key)]
        print(f"evaluated on {key}, calibrated on the other set: per channel matched {ch['matched_class_and_prior']} of {ch['oracle_dets']} (engine {ch['engine_dets']}), "
              f"IoU matched {ch['mask_iou_matched']}; per tensor matched {pt['matched_class_and_prior']} (engine {pt['engine_dets']}), IoU matched {pt['mask_iou_matched']}")
        assert ch["matched_class_and_prior"] >= pt["matched_class_and_prior"] - 2, (key, ch, pt)
        assert (ch["mask_iou_matched"] or 0) >= (pt["mask_iou_matched"] or 0) - 0.05, (key, ch, pt)
        assert ch["matched_class_and_prior"] >= 0.6 * ch["oracle_dets"], (key, ch)
