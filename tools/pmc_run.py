"""Small eager (no graph) forward for rocprofv3 --pmc passes: batch 16, two forwards."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = ya.Engine(input_size=550, max_batch=b, use_graph=False)
eng.load_weights(eng.generate_weights(1))
eng.set_input(np.random.default_rng(0).integers(0, 256, (b, 550, 550, 3), dtype=np.uint8))
for _ in range(2):
    eng.evaluate()
eng.sync()
