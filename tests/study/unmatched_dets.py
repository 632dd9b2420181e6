"""Study: which detections differ between the engine and the oracle's tail on the ENGINE's heads vs the ORACLE's heads,
for one slot of the seeded batch-64 frames (tests/test_gpu_fullsize.py). Usage: unmatched_dets.py [slot] [k=v tuning ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import yolact_amd as ya
import oracle
slot = int(sys.argv[1]) if len(sys.argv) > 1 else 37
tune = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:] if "=" in a}
S = 550
frames = np.random.default_rng(64).integers(0, 256, (64, S, S, 3), dtype=np.uint8)
eng = ya.Engine(input_size=S, max_batch=8, use_graph=True, tune=tune or None)
blob = eng.generate_weights(seed=1)
eng.load_weights(blob)
sub = frames[slot - slot % 8: slot - slot % 8 + 8]
eng.set_input(sub); eng.evaluate()
k = slot % 8
net = oracle.Net(50, S, 81, blob=blob)
want = net.forward(frames[slot:slot + 1], f16=True)
od, om = oracle.detect(want[0][0], want[1][0], want[2][0], want[3][0], net.priors())
ed, em = eng.detections(k)
key = lambda d: (d["class_id"], d["prior"])
ok, ek = {key(d): d for d in od}, {key(d): d for d in ed}
print("tuning", tune, "oracle dets", len(od), "engine dets", len(ed), "min score oracle", min(d["score"] for d in od), "engine", min(d["score"] for d in ed))
for name, a, b in (("oracle only", ok, ek), ("engine only", ek, ok)):
    for kk, d in a.items():
        if kk not in b:
            rank = sorted((x["score"] for x in a.values()), reverse=True).index(d["score"])
            print(f"  {name}: class {d['class_id']} prior {d['prior']} score {d['score']:.6f} rank {rank} box {[round(float(v), 4) for v in d['box']]}")
# the same tail on the ENGINE's heads: is the difference in the heads (numerics) or in the tail?
heads = [eng.output(i)[k] for i in range(4)]
od2, _ = oracle.detect(heads[0], heads[1], heads[2], heads[3], net.priors())
print("oracle tail on the engine's heads equals the engine's tail:", sorted(map(key, od2)) == sorted(map(key, ed)))
