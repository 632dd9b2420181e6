"""Wall-clock latency of the reference's own entry point: Yolact::classify on one 640x480 packed-u32
frame (src/yolact.rs:39; two 224x224 tiles), host buffer in, host buffer out, per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya

rng = np.random.default_rng(0)
frame0 = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
for mode, name in ((ya.COMPAT_SANE, "sane"), (ya.COMPAT_STRICT, "strict")):
    y = ya.Yolact.init(seed=1, compat_mode=mode)
    t = []
    for i in range(220):
        f = frame0.copy()
        t0 = time.perf_counter()
        try:
            y.classify(f)
        except ya.YhError as e:     # strict mode reports YH_EDIVERGE where the reference would hang
            print(name, "classify raised", e); break
        t.append(time.perf_counter() - t0)
    if t:
        t = np.array(t[20:]) * 1e3
        print(f"classify 640x480 ({name}): median {np.median(t):.3f} ms, p99 {np.percentile(t, 99):.3f} ms, {1e3 / np.median(t):.0f} frames/s")
