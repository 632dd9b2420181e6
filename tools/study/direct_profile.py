"""Per-launch times (event-bracketed, serialised) of a batch-N step with and without conv_direct_f16, side by side per op label."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8)
res, blob = {}, None
for name, tune in (("direct", dict(direct=1 << 20)), ("tiled", dict(direct=0))):   # (1 << 20: every eligible launch)
    e = ya.Engine(input_size=550, max_batch=batch, use_graph=True, tune=tune)
    if blob is None: blob = e.generate_weights(1)
    e.load_weights(blob); e.set_input(frames)
    for _ in range(3): e.evaluate()
    res[name] = e.profile(reps=20)
def key(n): return n.split(":")[-1].replace("/splitk", "")
tiled = {}
for r in res["tiled"]:
    tiled.setdefault(key(r["name"]), []).append(r)
for r in res["direct"]:
    if "conv_direct" not in r["name"]: continue
    t = tiled.get(key(r["name"]), [])
    print(f"{key(r['name']):12s} direct {r['ms'] * 1e3:7.2f} us   tiled {sum(x['ms'] for x in t) * 1e3:7.2f} us ({len(t)} launches)  {r['flops'] / 1e9:.2f} GFLOP")
