"""CPU study behind DESIGN.md §10's accuracy table: what each fp8 scaling scheme does to the
DETECTIONS of the configuration, measured with the oracle alone (no GPU) against the f16 oracle on the same frame.

    python tests/study/fp8_accuracy_table.py [--backbone 101 --size 700] [--frames noise,balls]

Schemes (oracle/orc_net.c, orc_net_set_fp8_study_ex): activations with one scale per tensor (what the engine ships), E8M0
blocks of 32 channels (what v_mfma_scale_* applies in hardware), one scale per input channel; weights per output channel
or in E8M0 blocks; and hybrids that keep named layers in f16."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", type=int, default=101)
    ap.add_argument("--size", type=int, default=700)
    ap.add_argument("--frames", default="noise,balls")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import bench
    import oracle as O
    net = O.Net(a.backbone, a.size, 81, seed=a.seed)
    pri = net.priors()
    frames = {}
    if "noise" in a.frames:
        import torch
        g = torch.Generator().manual_seed(0x594F4C41)
        frames["noise"] = torch.randint(0, 256, (1, a.size, a.size, 3), dtype=torch.uint8, generator=g).numpy()
    if "balls" in a.frames:
        frames["balls"] = bench.acceptance_frame(a.size)
    variants = [
        ("per-tensor act / per-channel w (shipped)", dict(act_mode=1, w_mode=1, skip="")),
        ("E8M0 blocks on activations", dict(act_mode=2, w_mode=1, skip="")),
        ("E8M0 blocks on both operands", dict(act_mode=2, w_mode=2, skip="")),
        ("per-input-channel act scales", dict(act_mode=3, w_mode=1, skip="")),
        ("head_t in f16", dict(act_mode=1, w_mode=1, skip="head_t")),
        ("head_t + proto3 in f16", dict(act_mode=1, w_mode=1, skip="head_t,proto3")),
        ("head_t + protonet in f16", dict(act_mode=1, w_mode=1, skip="head_t,proto")),
        ("head_t + protonet + FPN in f16 (backbone only)", dict(act_mode=1, w_mode=1, skip="head_t,proto,p")),
        ("backbone in f16 (FPN + protonet + head_t fp8)", dict(act_mode=1, w_mode=1, skip="l")),
    ]
    for fname, img in frames.items():
        t0 = time.time()
        h = net.forward(img, f16=True)
        ref = O.detect(h[0][0], h[1][0], h[2][0], h[3][0], pri)
        print(f"\n## frame '{fname}' {a.size}x{a.size}, R{a.backbone}: f16 oracle {len(ref[0])} detections ({time.time() - t0:.1f} s per forward)")
        print("| scheme | dets | matched (class, prior) | mask IoU all | mask IoU matched | conf rms % | proto rms % |")
        print("|---|---|---|---|---|---|---|")
        for label, kw in variants:
            net.set_fp8_study_ex(**kw)
            q = net.forward(img, f16=True)
            net.set_fp8_study_ex(0, 1, "")
            dq = O.detect(q[0][0], q[1][0], q[2][0], q[3][0], pri)
            acc = bench.accuracy_vs_oracle(dq, ref)
            rms = lambda i: 100 * float(np.sqrt(((q[i] - h[i]) ** 2).mean()) / np.sqrt((h[i] ** 2).mean()))
            print(f"| {label} | {acc['engine_dets']} | {acc['matched_class_and_prior']} / {acc['oracle_dets']} | {acc['mask_iou_all']} | {acc['mask_iou_matched']} | {rms(1):.2f} | {rms(3):.2f} |", flush=True)


if __name__ == "__main__":
    main()
