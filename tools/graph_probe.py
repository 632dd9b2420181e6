"""Graph-replay probe for the profiler crash study: replays the YOLACT engine's captured step
as a SINGLE-BRANCH graph (yh_invoke: no detection tail, hence no fork onto the side stream) or with the tail
(yh_evaluate: forked graph). Usage: graph_probe.py [invoke|evaluate] [replays]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
mode = sys.argv[1] if len(sys.argv) > 1 else "invoke"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = ya.Engine(input_size=224, max_batch=2, use_graph=True)
eng.load_weights(eng.generate_weights(1))
eng.set_input(np.random.default_rng(0).integers(0, 256, (2, 224, 224, 3), dtype=np.uint8))
tail = mode == "evaluate"
for _ in range(3):
    (eng.evaluate if tail else eng.invoke)()
eng.sync()
ms = eng.time_steps(reps, tail) / reps
print(f"{mode}: {reps} graph replays, {ms:.4f} ms per step", flush=True)
