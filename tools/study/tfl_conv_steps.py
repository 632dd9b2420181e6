"""Device time of ONE int8-MFMA CONV_2D launch as a function of its k-steps (kernel 1 / 3 / 5 x Ci 64..256) and pixels: fixed cost vs cost per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("tiny-object-detection_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import yolact_amd as ya
import tfl_builder as B, tfl_models as M
rng = np.random.default_rng(0)
dot = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # yh_tuning.tfl_dot: 2 = 64 x 64 tiles through LDS, 3 = register-fed 32 x 32 wave tiles
print(f"tfl_dot = {dot}")
for hw in (4, 14, 28, 56):
    for k, ci in ((1, 64), (1, 128), (1, 256), (3, 64), (3, 128), (5, 128)):
        m = M.single_op("CONV_2D", rng, k=k, stride=1, padding=0, act=1, h=hw, w=hw, ci=ci, co=128, so=0.3)
        e = ya.TfliteEngine(bytes(B.serialize(m)), tune=dict(tfl_dot=dot))
        x = rng.integers(0, 256, (1, hw, hw, ci), dtype=np.uint8)
        e.set_input(x)
        for _ in range(50): e.invoke()
        e.output(0)
        N = 2000
        t0 = time.perf_counter()
        for _ in range(N): e.invoke()
        e.output(0)
        dt = (time.perf_counter() - t0) / N * 1e6
        steps = k * k * ci // 64
        print(f"{hw:3d}x{hw:<3d} k={k} Ci={ci:3d}: {steps:3d} k-steps, {-(-hw * hw // 64):3d} x 2 workgroups: {dt:6.2f} us per launch", flush=True)
        e.close()
