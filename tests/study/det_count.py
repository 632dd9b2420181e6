import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/study -> repo root
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from PIL import Image
S = int(sys.argv[1])
for name in ("frc_balls", "red_robot", "noise"):
    if name == "noise": img = np.random.default_rng(0).integers(0, 256, (1, S, S, 3), dtype=np.uint8)
    else: img = np.asarray(Image.open(os.path.join(ROOT, f"tests/golden/{name}.png")).convert("RGB").resize((S, S), Image.BILINEAR))[None]
    net = O.Net(50, S, 81, seed=1)
    a = net.forward(img, f16=True)
    for th in (0.05, 0.02, 0.01, 0.005):
        d, _ = O.detect(a[0][0], a[1][0], a[2][0], a[3][0], net.priors(), conf_thresh=th, want_masks=False)
        print(S, name, th, len(d))
