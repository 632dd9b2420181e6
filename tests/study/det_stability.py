"""CPU study: how stable are the oracle's detections under storage-precision noise (f16 vs f32 layers)?
Proxy for engine-vs-oracle summation-order noise. Usage: python tools/exp/det_stability.py [size]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/study -> repo root
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from PIL import Image
S = int(sys.argv[1]) if len(sys.argv) > 1 else 550
img = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/frc_balls.png")).convert("RGB").resize((S, S), Image.BILINEAR))[None]
if os.environ.get("NOISE"): img = np.random.default_rng(int(os.environ["NOISE"])).integers(0,256,(1,S,S,3),dtype=np.uint8)
net = O.Net(50, S, 81, seed=1)
pri = net.priors()
t = time.time()
a = net.forward(img, f16=True)
b = net.forward(img, f16=False)
print("fwd s", time.time() - t)
da, ma = O.detect(a[0][0], a[1][0], a[2][0], a[3][0], pri)
db, mb = O.detect(b[0][0], b[1][0], b[2][0], b[3][0], pri)
ka = {(d["class_id"], d["prior"]) for d in da}
kb = {(d["class_id"], d["prior"]) for d in db}
print("dets", len(da), len(db), "matched", len(ka & kb))
sc = [d["score"] for d in da]
print("scores: max %.4f median %.4f min %.4f" % (max(sc), np.median(sc), min(sc)))
def union_iou(d1, m1, d2, m2):
    inter = union = 0
    for c in set(d["class_id"] for d in d1) | set(d["class_id"] for d in d2):
        u1 = np.zeros(m1.shape[1:], bool); u2 = np.zeros(m1.shape[1:], bool)
        for i, d in enumerate(d1):
            if d["class_id"] == c: u1 |= m1[i] > 0
        for i, d in enumerate(d2):
            if d["class_id"] == c: u2 |= m2[i] > 0
        inter += (u1 & u2).sum(); union += (u1 | u2).sum()
    return inter / max(union, 1)
print("per-class union-of-masks IoU", union_iou(da, ma, db, mb))
kb2 = {(d["class_id"], d["prior"]): j for j, d in enumerate(db)}
i_ = u_ = 0
for i, d in enumerate(da):
    j = kb2.get((d["class_id"], d["prior"]))
    if j is not None: i_ += (ma[i] & mb[j]).sum(); u_ += (ma[i] | mb[j]).sum()
print("matched IoU", i_ / max(u_, 1), "mask fill", float(ma.mean()) if len(da) else 0)
# candidates above threshold
conf = a[1][0]; z = conf - conf.max(1, keepdims=True); p = np.exp(z); p /= p.sum(1, keepdims=True)
print("candidates", int((p[:, 1:] > 0.05).sum()), "classes with dets", len(set(d["class_id"] for d in da)))
