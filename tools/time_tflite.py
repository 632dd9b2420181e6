"""The reference's own model family on the GPU: a full-size synthetic MobileNetV2-FPN-YOLACT .tflite
(tests/tfl_models.mobilenetv2_yolact: the op census of data/FRC_model_edgetpu.log; the real
FRC_model.tflite is absent) through the uint8 executor: per-invoke latency (host input copy + graph
launch + sync) and classify() on a 640x480 frame. Parity of this model against the numpy oracle is
tests/test_gpu_tflite.py::test_full_size_mobilenetv2_yolact_graph (tools/ never load oracle/)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tiny-object-detection_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import yolact_amd as ya
import tfl_builder as B, tfl_models as M

ap = argparse.ArgumentParser()
ap.add_argument("--graph", type=int, default=-1, help="yh_tuning.tfl_graph: 1 = hipGraph replay of the plan, 0 = eager launches, -1 = library default")
ap.add_argument("--invokes", type=int, default=220)
a = ap.parse_args()
rng = np.random.default_rng(0)
model = M.mobilenetv2_yolact(rng)
buf = bytes(B.serialize(model))
eng = ya.TfliteEngine(buf, tune=dict(tfl_graph=a.graph))
x = rng.integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
def invoke():
    eng.set_input(x); eng.invoke(); return eng.output(4)
got = invoke()
print(f"{len(model.ops)} ops, {len(buf) / 1e6:.1f} MB model; output 4: {got.shape}, {len(np.unique(got))} distinct codes")
t = []
for _ in range(a.invokes):
    t0 = time.perf_counter(); invoke(); t.append(time.perf_counter() - t0)
t = np.array(t[20:]) * 1e3
print(f"tfl_graph={a.graph}; invoke 224x224 (set_input + invoke + read output 4): median {np.median(t):.3f} ms, p99 {np.percentile(t, 99):.3f} ms")
frame = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
t = []
for _ in range(120):
    f = frame.copy(); t0 = time.perf_counter(); eng.classify_frame(f, 640, 480, ya.COMPAT_SANE); t.append(time.perf_counter() - t0)
t = np.array(t[20:]) * 1e3
print(f"classify 640x480 through the .tflite (two tiles): median {np.median(t):.3f} ms, p99 {np.percentile(t, 99):.3f} ms")
