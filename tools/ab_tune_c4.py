#!/usr/bin/env python3
"""ab_tune.py for configs[4]'s per-GPU share (YOLACT-700 R101, fp8 precision, 8 frames): interleaved A/B of yh_tuning variants,
median / min of graph-replayed steps, and equality of every output with the first variant's.
usage: ab_tune_c4.py [BATCH] variant [variant ...]    variant = "-" (defaults) or "field=value[,field=value...]" """
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

args = sys.argv[1:]
batch = int(args.pop(0)) if args and args[0].isdigit() else 8
variants = args or ["-"]
frames = np.random.default_rng(0).integers(0, 256, (batch, 700, 700, 3), dtype=np.uint8)
engs, blob, scales = {}, None, None
for v in variants:
    tune = {} if v == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")}
    e = ya.Engine(input_size=700, backbone=101, max_batch=batch, use_graph=True, precision=ya.PRECISION_FP8, tune=tune)
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob)
    e.set_input(frames)
    if scales is None:
        e.fp8_calibrate()
        scales = e.fp8_channel_scales()
    else:
        for i, (_, sc) in enumerate(scales):
            e.fp8_set_layer_scale(i, sc)
    for _ in range(3):
        e.evaluate()
    e.sync()
    engs[v] = e
ref = [engs[variants[0]].output(i) for i in range(4)]
for v in variants[1:]:
    same = all(np.array_equal(engs[v].output(i), ref[i]) for i in range(4))
    print(f"{v}: outputs {'bit-equal to' if same else 'DIFFER from'} {variants[0]}", flush=True)
steps, rounds = 30, 8
t = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        t[v].append(engs[v].time_steps(steps, True) / steps)
for v in variants:
    a = np.array(t[v])
    print(f"R101-700 fp8 batch {batch} {v:32s}: median {np.median(a):.4f} ms/step, min {a.min():.4f}  -> {batch / np.median(a) * 1e3:.1f} frames/s", flush=True)
