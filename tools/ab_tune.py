#!/usr/bin/env python3
"""Interleaved A/B of yh_tuning variants in ONE process (guide rule 24): median / min of graph-replayed steps.
usage: ab_tune.py BATCH[@SIZE] variant [variant ...]    variant = "-" (defaults) or "field=value[,field=value...]"; SIZE = input size (default 550; 224 = the tiles of the reference's classify) """
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

batch, size = (int(x) for x in (sys.argv[1] + "@550").split("@")[:2])
variants = sys.argv[2:] or ["-"]
frames = np.random.default_rng(0).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)
engs, blob = {}, None
for v in variants:
    tune = {} if v == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")}
    e = ya.Engine(input_size=size, max_batch=batch, use_graph=True, tune=tune)
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob)
    e.set_input(frames)
    for _ in range(3):
        e.evaluate()
    e.sync()
    engs[v] = e
steps, rounds = (10, 8) if batch >= 32 else (100, 8)
t = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        t[v].append(engs[v].time_steps(steps, True) / steps)
for v in variants:
    a = np.array(t[v])
    print(f"batch {batch} @{size} {v:40s}: median {np.median(a):.4f} ms/step, min {a.min():.4f}  -> {batch / np.median(a) * 1e3:.1f} frames/s", flush=True)
