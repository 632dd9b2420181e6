"""Per-launch hipEvent profile of one forward+tail (yh_profile_run), printed as a table."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=550)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--top", type=int, default=200)
ap.add_argument("--backbone", type=int, default=50)
ap.add_argument("--precision", default="f16", choices=("f16", "fp8"))
ap.add_argument("--tune", default="", help="comma-separated yh_tuning fields, e.g. tailfork=0,k1tile=0")
a = ap.parse_args()
tune = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
eng = ya.Engine(input_size=a.size, backbone=a.backbone, max_batch=a.batch, use_graph=False, tune=tune,
                precision=ya.PRECISION_FP8 if a.precision == "fp8" else ya.PRECISION_F16)
eng.load_weights(eng.generate_weights(1))
rng = np.random.default_rng(0)
eng.set_input(rng.integers(0, 256, (a.batch, a.size, a.size, 3), dtype=np.uint8))
if a.precision == "fp8":
    eng.fp8_calibrate()
eng.evaluate(); eng.sync()
prof = eng.profile(True, a.reps)
tot = sum(p["ms"] for p in prof)
print(f"YOLACT-{a.size} R{a.backbone} {a.precision} batch {a.batch}: total {tot:.3f} ms over {len(prof)} launches -> {a.batch / tot * 1e3:.1f} fps (serialised, event-bracketed)")
by = {}
for p in prof:
    d = by.setdefault(p["name"].split(":")[0], [0.0, 0.0, 0.0, 0])
    d[0] += p["ms"]; d[1] += p["flops"]; d[2] += p["bytes"]; d[3] += 1
print("--- by kernel")
for k, d in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:40s} n={d[3]:3d} {d[0]:9.3f} ms {100 * d[0] / tot:5.1f}%  {d[1] / max(d[0], 1e-9) / 1e9:8.1f} TFLOP/s  {d[2] / max(d[0], 1e-9) / 1e6:8.1f} GB/s")
print("--- by launch")
for p in sorted(prof, key=lambda p: -p["ms"])[:a.top]:
    print(f"{p['name']:52s} {p['ms']:8.4f} ms  {p['flops'] / max(p['ms'], 1e-9) / 1e9:8.1f} TFLOP/s  {p['bytes'] / max(p['ms'], 1e-9) / 1e6:8.1f} GB/s  {p['flops'] / 1e9:8.2f} GFLOP")
