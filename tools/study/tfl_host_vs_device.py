"""Where does a .tflite invoke spend its wall time? Host time inside set_input / invoke (enqueue) / output read, per plan."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("tiny-object-detection_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import yolact_amd as ya
import tfl_builder as B, tfl_models as M
rng = np.random.default_rng(0)
model = M.mobilenetv2_yolact(rng)
buf = bytes(B.serialize(model))
x = rng.integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
for v in (dict(tfl_fuse=0), dict(tfl_fuse=1), dict(tfl_fuse=1, tfl_graph=1)):
    e = ya.TfliteEngine(buf, tune=v)
    for _ in range(20):
        e.set_input(x); e.invoke(); e.output(4)
    a = b = c = 0.0
    N = 200
    for _ in range(N):
        t0 = time.perf_counter(); e.set_input(x); t1 = time.perf_counter(); e.invoke(); t2 = time.perf_counter(); e.output(4); t3 = time.perf_counter()
        a += t1 - t0; b += t2 - t1; c += t3 - t2
    # back-to-back invokes without reading (device throughput if the host keeps up)
    t0 = time.perf_counter()
    for _ in range(N):
        e.invoke()
    e.output(4)
    tb = (time.perf_counter() - t0) / N
    print(f"{v}: set_input {a / N * 1e3:.3f} ms, invoke (enqueue) {b / N * 1e3:.3f} ms, output read (wait + copy) {c / N * 1e3:.3f} ms; back-to-back invokes {tb * 1e3:.3f} ms each", flush=True)
    e.close()
