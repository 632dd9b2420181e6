"""Layer 3's expand + next-reduce launch, every form, timed per launch (hipEvent-bracketed profile run) and per step (graph replay):
usage: xn_forms.py BATCH [size backbone precision]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 550
bb = int(sys.argv[3]) if len(sys.argv) > 3 else 50
prec = ya.PRECISION_FP8 if len(sys.argv) > 4 and sys.argv[4] == "fp8" else ya.PRECISION_F16
frames = np.random.default_rng(0).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)
blob, engs = None, []
for name, tune in (("separate launches", dict(chain=17 + 128)), ("two-phase, 64 px", dict(xn_tm=64)), ("two-phase, 128 px", dict(xn_tm=128)),
                   ("pipelined, 64 px", dict(xn_tm=64, xn_pipe=1)), ("pipelined, 128 px", dict(xn_tm=128, xn_pipe=1))):
    e = ya.Engine(input_size=size, backbone=bb, max_batch=batch, use_graph=True, precision=prec, tune=tune)
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob); e.set_input(frames)
    if prec == ya.PRECISION_FP8:
        e.fp8_calibrate()
    for _ in range(3): e.evaluate()
    e.sync()
    prof = e.profile(True, 3)
    ms = [p["ms"] for p in prof if p["name"].startswith("bneck_xn")]
    sep = [p["ms"] for p in prof if p["name"].split(":")[1] in ("l3b1", "l3b2_a")] if not ms else []
    engs.append((name, e, ms, sep))
steps, rounds = (10, 6) if batch >= 32 else (40, 6)
t = {n: [] for n, *_ in engs}
for r in range(rounds):
    for n, e, *_ in engs:
        t[n].append(e.time_steps(steps, True) / steps)
for n, e, ms, sep in engs:
    per = f"{len(ms)} launches, {1e3 * np.mean(ms):6.1f} us each" if ms else f"l3b1 + l3b2_a: {1e3 * sum(sep):6.1f} us"
    print(f"batch {batch} {size} R{bb} {n:20s}: step median {np.median(t[n]):.4f} ms (min {min(t[n]):.4f}); {per}", flush=True)
