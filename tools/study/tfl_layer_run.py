"""Driver for tfl_layer_table.py: N batch-2 invokes (the two tiles of a classify) of the 136-op stand-in, nothing else on the device.
usage: tfl_layer_run.py [field=value,...]   (yh_tuning fields)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("tiny-object-detection_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import yolact_amd as ya
import tfl_builder as B, tfl_models as M
tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[1].split(",")} if len(sys.argv) > 1 else {}
rng = np.random.default_rng(0)
e = ya.TfliteEngine(bytes(B.serialize(M.mobilenetv2_yolact(rng))), tune=tune)
x = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)
e.set_batch(2)
e.set_input(x)
ps = e.plan_summary()
print(ps, flush=True)
print("LAUNCHES", ps["launches_per_invoke"], flush=True)
for _ in range(300):
    e.invoke()
e.output(4)
