"""Graph-replayed step time (yh_time_steps: one hipEvent pair on the engine's stream)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
# usage: time_steps.py [batch ...] [field=value ...]   (yh_tuning fields, e.g. tailfork=0)
tune = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
for b in [int(x) for x in sys.argv[1:] if "=" not in x] or [1, 64]:
    eng = ya.Engine(input_size=550, max_batch=b, use_graph=True, tune=tune)
    eng.load_weights(eng.generate_weights(1))
    eng.set_input(np.random.default_rng(0).integers(0, 256, (b, 550, 550, 3), dtype=np.uint8))
    for _ in range(3):
        eng.evaluate()
    eng.sync()
    steps = 50 if b <= 8 else 10
    ms = eng.time_steps(steps, True) / steps
    print(f"batch {b} {tune or ''}: {ms:.4f} ms/step -> {b / ms * 1e3:.1f} fps (graph replay, device time)", flush=True)
    eng.close()
