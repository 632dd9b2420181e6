#!/usr/bin/env python3
"""Interleaved A/B of yh_tuning.chain variants in ONE process (guide rule 24): median / min of graph-replayed batch-64 steps.
usage: ab_chain.py [batch] [chain values ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 65]
frames = np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8)
engs = {}
blob = None
for v in variants:
    e = ya.Engine(input_size=550, max_batch=batch, use_graph=True, tune=dict(chain=v))
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob)
    e.set_input(frames)
    for _ in range(3):
        e.evaluate()
    e.sync()
    engs[v] = e
steps, rounds = (10, 8) if batch >= 32 else (50, 8)
t = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        t[v].append(engs[v].time_steps(steps, True) / steps)
for v in variants:
    a = np.array(t[v])
    desc = "off" if not v & 1 else ("on" + (", 128-px tile for 64 planes" if v & 2 else "") + (", persistent grid" if v & 4 else ", one WG per tile") +
                                    (f", stagger {(v >> 8) * 64 if v >> 8 else 704} x 64 clk" if v & 8 else "") + (", layer 1 only" if v & 32 else "") + (", no first-block form" if v & 64 else ""))
    print(f"chain={v:5d} ({desc}): median {np.median(a):.4f} ms/step, min {a.min():.4f}  -> {batch / np.median(a) * 1e3:.1f} frames/s", flush=True)
