"""configs[4] (YOLACT-700 R101 fp8, 8 frames) with and without the layer-3 expand + next-reduce launch, interleaved."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
for batch in (8, 12):
    frames = np.random.default_rng(0).integers(0, 256, (batch, 700, 700, 3), dtype=np.uint8)
    engs, blob = [], None
    for label, tune, prec in (("fp8, xn on", {}, ya.PRECISION_FP8), ("fp8, xn off", {"chain": 145}, ya.PRECISION_FP8), ("f16, xn on", {}, ya.PRECISION_F16), ("f16, xn off", {"chain": 145}, ya.PRECISION_F16)):
        e = ya.Engine(input_size=700, backbone=101, max_batch=batch, use_graph=True, precision=prec, tune=tune or None)
        if blob is None: blob = e.generate_weights(1)
        e.load_weights(blob); e.set_input(frames)
        if prec == ya.PRECISION_FP8: e.fp8_calibrate()
        for _ in range(3): e.evaluate()
        e.sync(); engs.append((label, e))
    t = {l: [] for l, _ in engs}
    for _ in range(6):
        for l, e in engs: t[l].append(e.time_steps(20, True) / 20)
    for l, e in engs:
        ms = float(np.median(t[l])); print(f"batch {batch:2d} {l:14s} {ms:8.4f} ms/step {batch / ms * 1e3:8.1f} frames/s", flush=True); e.close()
