"""GPU study behind the detection-level fp8 assertion: how far apart are the fp8 ENGINE and the fp8-mode
ORACLE at the level of detections, and which detections are decided by more than the head noise between them?

    python tests/study/fp8_det_margin.py [--backbone 101 --size 700 --frames 2]

Prints, per frame: head rms differences, the probability ratio engine / oracle for every oracle detection, and matched
fractions as a function of the score margin above the cut."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def softmax_prob(conf_row, cls):
    z = conf_row.astype(np.float64)
    pe = np.exp(z - z.max())
    return float(pe[cls + 1] / pe.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", type=int, default=101)
    ap.add_argument("--size", type=int, default=700)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--skip", default="")
    a = ap.parse_args()
    import bench
    import oracle as O
    import yolact_amd as ya
    S = a.size
    eng = ya.Engine(input_size=S, backbone=a.backbone, max_batch=a.frames, use_graph=True, precision=ya.PRECISION_FP8)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    rng = np.random.default_rng(23)
    frames = rng.integers(0, 256, (a.frames, S, S, 3), dtype=np.uint8)
    acc = bench.acceptance_frame(S)
    if acc is not None:
        frames[-1] = acc[0]
    eng.set_input(frames)
    eng.fp8_calibrate()
    eng.evaluate()
    heads = [eng.output(i) for i in range(4)]
    net = O.Net(a.backbone, S, 81, blob=blob)
    lay = {}
    for name, sc in eng.fp8_channel_scales():
        for nm in ([f"{name}{l}" for l in range(5)] if name == "head_t" else [name]):
            lay[nm] = sc
    pri = net.priors()
    for f in range(a.frames):
        net.set_fp8(lay)
        w8 = net.forward(frames[f:f + 1], f16=True)
        net.set_fp8(None)
        w16 = net.forward(frames[f:f + 1], f16=True)
        ed = eng.detections(f)
        for tag, want in (("fp8 oracle", w8), ("f16 oracle", w16)):
            od = O.detect(want[0][0], want[1][0], want[2][0], want[3][0], pri)
            print(f"\n=== frame {f} ({'frc_balls' if acc is not None and f == a.frames - 1 else 'noise'}): engine (fp8) vs {tag}")
            for i, tn in enumerate(("loc", "conf", "mask", "proto")):
                d = heads[i][f] - want[i][0]
                print(f"  {tn}: rms diff {100 * float(np.sqrt((d ** 2).mean()) / np.sqrt((want[i][0] ** 2).mean())):.2f} % of rms, max {float(np.abs(d).max()):.3f}")
            acc_ = bench.accuracy_vs_oracle(ed, od)
            print("  ", {k: acc_[k] for k in ("oracle_dets", "engine_dets", "matched_class_and_prior", "mask_iou_all", "mask_iou_matched")})
            # probability continuity: for every oracle detection, the engine's own probability of that (class, prior)
            ratio = np.array([softmax_prob(heads[1][f][d["prior"]], d["class_id"]) / d["score"] for d in od[0]])
            lr = np.abs(np.log(ratio))
            print(f"   engine prob / oracle score over the oracle's {len(ratio)} detections: median |log ratio| {np.median(lr):.3f}, 90 % {np.quantile(lr, 0.9):.3f}, max {lr.max():.3f}")
            if len(od[0]) == 100 and len(ed[0]) == 100:
                cut = max(od[0][-1]["score"], ed[0][-1]["score"])
                ek = {(d["class_id"], d["prior"]) for d in ed[0]}
                ok_ = {(d["class_id"], d["prior"]) for d in od[0]}
                for m in (1.0, 1.1, 1.25, 1.5, 2.0, 3.0):
                    oo = [d for d in od[0] if d["score"] > cut * m]
                    ee = [d for d in ed[0] if d["score"] > cut * m]
                    mo = sum((d["class_id"], d["prior"]) in ek for d in oo)
                    me = sum((d["class_id"], d["prior"]) in ok_ for d in ee)
                    print(f"   margin x{m}: oracle dets above {len(oo)}, matched in engine {mo}; engine dets above {len(ee)}, matched in oracle {me}")
            # masks of matched pairs: per-pair IoU distribution
            ekm = {(d["class_id"], d["prior"]): i for i, d in enumerate(ed[0])}
            ious = []
            for j, d in enumerate(od[0]):
                i = ekm.get((d["class_id"], d["prior"]))
                if i is None:
                    continue
                x, y = ed[1][i] > 0, od[1][j] > 0
                u = int((x | y).sum())
                if u:
                    ious.append((int((x & y).sum()) / u, u))
            if ious:
                v = np.array([t[0] for t in ious]); u = np.array([t[1] for t in ious])
                print(f"   matched pairs with a mask: {len(v)}; per-pair IoU median {np.median(v):.3f}, 10 % {np.quantile(v, 0.1):.3f}; union pixels median {int(np.median(u))}; pixel-weighted {float((v * u).sum() / u.sum()):.3f}")
    eng.close()


if __name__ == "__main__":
    main()
