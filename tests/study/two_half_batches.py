"""Study (not a test): does a batch-64 step run faster as TWO concurrent half-batch engines (32 frames each, own streams)?
Complementary phases (one half in an MFMA-bound 3x3 conv while the other streams a 1x1) could overlap. Prints ms per 64 frames."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total = 64
rng = np.random.default_rng(0)
engs = []
for k in range(parts):
    e = ya.Engine(input_size=550, max_batch=total // parts, use_graph=True)
    e.load_weights(e.generate_weights(1))
    e.set_input(rng.integers(0, 256, (total // parts, 550, 550, 3), dtype=np.uint8))
    engs.append(e)
for _ in range(3):
    for e in engs: e.evaluate()
for e in engs: e.sync()
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    for e in engs: e.evaluate()
for e in engs: e.sync()
dt = (time.perf_counter() - t0) / steps * 1e3
print(f"{parts} engine(s) x batch {total // parts}: {dt:.3f} ms per {total} frames -> {total / dt * 1e3:.0f} fps (wall clock, graph replay)", flush=True)
for e in engs: e.close()
