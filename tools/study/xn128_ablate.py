"""Timing study of bneck_xn128_f16 (batch 64): which stream binds it? tune.ablate bits 4-7 drop the residual / y-store / W_c / W_a' stream."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8)
blob = None
for name, ab in (("all streams", 0), ("no residual", 16), ("no y stores", 32), ("no W_c", 64), ("no W_a'", 128), ("no weights", 192), ("no res, no stores", 48), ("nothing", 240)):
    e = ya.Engine(input_size=550, max_batch=batch, use_graph=False, tune=dict(ablate=ab, xn_tm=128))
    if blob is None:
        blob = e.generate_weights(1)
    e.load_weights(blob); e.set_input(frames); e.evaluate(); e.sync()
    prof = e.profile(True, 3)
    ms = [p["ms"] for p in prof if p["name"].startswith("bneck_xn128")]
    print(f"{name:20s}: {len(ms)} launches, mean {1e3 * np.mean(ms):7.1f} us  ({', '.join(f'{1e3*m:.0f}' for m in ms)})", flush=True)
    e.close()
