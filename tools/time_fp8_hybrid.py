#!/usr/bin/env python3
"""configs[4] (YOLACT-700 R101, fp8) with groups of its E4M3 layers kept in f16 (yh_config.fp8_f16_layers): step time at 8 frames per
GPU (configs[4]'s per-GPU share) and at batch 64, interleaved in one process - the frames/s column of DESIGN.md §10's table."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

VARIANTS = [("f16 engine", None), ("all 36 layers fp8", 0), ("head_t in f16", 1), ("head_t + protonet in f16", 3), ("head_t + protonet + FPN in f16 (backbone fp8)", 7),
            ("backbone in f16 (FPN + protonet + head_t fp8)", 8)]
for batch in (8, 64):
    frames = np.random.default_rng(0).integers(0, 256, (batch, 700, 700, 3), dtype=np.uint8)
    engs, blob = [], None
    for label, keep in VARIANTS:
        e = ya.Engine(input_size=700, backbone=101, max_batch=batch, use_graph=True,
                      precision=ya.PRECISION_F16 if keep is None else ya.PRECISION_FP8, fp8_f16_layers=keep or 0)
        if blob is None:
            blob = e.generate_weights(1)
        e.load_weights(blob)
        e.set_input(frames)
        if keep is not None:
            e.fp8_calibrate()
        for _ in range(3):
            e.evaluate()
        e.sync()
        engs.append((label, e, len(e.fp8_layers()) if keep is not None else 0))
    steps, rounds = (20, 6) if batch == 8 else (5, 5)
    t = {label: [] for label, _, _ in engs}
    for _ in range(rounds):
        for label, e, _ in engs:
            t[label].append(e.time_steps(steps, True) / steps)
    for label, e, nl in engs:
        ms = float(np.median(t[label]))
        print(f"batch {batch:2d}  {label:50s} {nl:2d} E4M3 launches  {ms:8.4f} ms/step  {batch / ms * 1e3:8.1f} frames/s", flush=True)
        e.close()
