"""First-contact GPU check: conv kernel vs oracle on a few shapes, then smoke()."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g._paths()
import yolact_amd as ya
import oracle as O

def f16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)

eng = ya.Engine(input_size=128, max_batch=2, use_graph=False)
print(ya.version(), flush=True)
rng = np.random.default_rng(0)
cases = [  # n,h,w,cin,cout,k,stride,pad,res,act
    (1, 8, 8, 64, 128, 1, 1, 0, False, 0),
    (1, 9, 7, 64, 64, 3, 1, 1, False, 1),
    (2, 17, 13, 128, 256, 3, 2, 1, True, 1),
    (1, 20, 20, 256, 32, 1, 1, 0, False, 1),
    (1, 12, 12, 256, 351, 3, 1, 1, False, 2),
    (1, 33, 31, 3, 64, 7, 2, 3, False, 1),
    (2, 16, 16, 256, 256, 3, 1, 1, False, 1),
]
bad = 0
for (n, h, w, cin, cout, k, s, p, res, act) in cases:
    x = f16(rng.normal(0, 1, (n, h, w, cin)))
    wt = f16(rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin))
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    r = f16(rng.normal(0, 1, (n, ho, wo, cout))) if res else None
    try:
        y = eng.op_conv2d(x, wt, b, s, p, r, act)
        yo = O.conv2d(x, wt, b, s, p, r, act, f16=True)
        err = np.abs(y - yo).max()
        print(f"conv {n}x{h}x{w}x{cin}->{cout} k{k}s{s} res={res} act={act}: maxerr {err:.5f} (ref absmax {np.abs(yo).max():.3f})", flush=True)
        if not (err < 0.02): bad += 1
    except Exception:
        traceback.print_exc(); bad += 1
print("bad", bad, flush=True)
try:
    g.smoke()
except Exception:
    traceback.print_exc(); bad += 1
sys.exit(1 if bad else 0)
