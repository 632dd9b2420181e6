"""The scene back-end (yh_scene_*: shaders/pt_cloud.comp + pt_cloud_weights.comp as HIP kernels) on one 640x480 frame:
device milliseconds per frame (five launches), for a terrain-only frame and one with robots and balls."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
H, W = 480, 640
rng = np.random.default_rng(0)
depth = rng.integers(200, 4000, (H, W)).astype(np.uint16)
sc = ya.Scene(W, H)
for name, robots in (("terrain only", False), ("robots + balls", True)):
    ci = np.zeros((H, W, 2), np.uint8)
    if robots:
        ci[100:220, 150:330, 0] = 1; ci[260:330, 380:520, 0] = 2; ci[60:75, 60:80] = (3, 4); ci[400:420, 500:530] = (3, 9)
    sc.append(depth, ci, ya.COMPAT_SANE)
    out = sc.read()
    print(f"scene 640x480, {name}: {sc.time(30):.3f} ms per frame (map max {out['map'].max()}, {int((out['balls'][:, 2] > 0).sum())} balls)")
