#!/usr/bin/env python3
"""Do two engines on ONE device overlap each other's phases? (a step is an HBM-bound backbone phase followed by an
MFMA-bound protonet / head phase.) Each engine owns its streams and graphs; the steps of the two are enqueued alternately from
one thread, optionally with the second engine started half a step late. usage: two_engines.py [batch] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames = np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8)
engs = []
blob = None
for i in range(2):
    e = ya.Engine(input_size=550, max_batch=batch, use_graph=True)
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob)
    e.set_input(frames)
    for _ in range(3):
        e.evaluate()
    e.sync()
    engs.append(e)


def run(n_eng, n, stagger_s=0.0):
    for e in engs[:n_eng]:
        e.sync()
    t0 = time.perf_counter()
    if n_eng == 2 and stagger_s > 0:
        engs[0].evaluate()
        time.sleep(stagger_s)
        for _ in range(n - 1):
            engs[1].evaluate(); engs[0].evaluate()
        engs[1].evaluate()
    else:
        for _ in range(n):
            for e in engs[:n_eng]:
                e.evaluate()
    for e in engs[:n_eng]:
        e.sync()
    return time.perf_counter() - t0


one = min(run(1, steps) for _ in range(3))
print(f"one engine : {steps} steps of {batch} frames in {one * 1e3:.2f} ms -> {steps * batch / one:.1f} frames/s", flush=True)
for st in (0.0, 0.003, 0.005, 0.007):
    two = min(run(2, steps, st) for _ in range(3))
    print(f"two engines (second started {st * 1e3:.0f} ms late): 2 x {steps} steps in {two * 1e3:.2f} ms -> {2 * steps * batch / two:.1f} frames/s ({2 * one / two:.3f} x one engine)", flush=True)
