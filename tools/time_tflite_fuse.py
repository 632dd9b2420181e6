"""TFLite executor: fused plan (yh_tuning.tfl_fuse = 1, default) against one launch per operator (0), eager and as a captured graph,
interleaved in one process: median ms per invoke (set_input + invoke + read output 4) and per classify(640x480)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tiny-object-detection_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import yolact_amd as ya
import tfl_builder as B, tfl_models as M
rng = np.random.default_rng(0)
model = M.mobilenetv2_yolact(rng)
buf = bytes(B.serialize(model))
x = rng.integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
frame = (rng.integers(0, 256, (480, 640, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
only = sys.argv[1:]   # e.g. "fuse=1,graph=0": run just that variant (for rocprofv3 --stats)
variants = [dict(tfl_fuse=f, tfl_graph=g, tfl_dot=dot, tfl_group=gr) for f, g, dot, gr in ((1, 0, 2, 0), (1, 0, 3, 0), (1, 0, 3, 1), (0, 0, 3, 1), (1, 1, 3, 1))]
if only:
    kv = dict(p.split("=") for p in only[0].split(","))
    variants = [dict(tfl_fuse=int(kv["fuse"]), tfl_graph=int(kv["graph"]), tfl_dot=int(kv.get("dot", 3)), tfl_group=int(kv.get("group", 1)))]
engs = [(v, ya.TfliteEngine(buf, tune=v)) for v in variants]
for v, e in engs:
    print(v, e.plan_summary(), flush=True)
def invoke(e):
    e.set_input(x); e.invoke(); return e.output(4)
res = {i: ([], []) for i in range(len(engs))}
for rnd in range(6):
    for i, (v, e) in enumerate(engs):
        for _ in range(10): invoke(e)
        t = []
        for _ in range(60):
            t0 = time.perf_counter(); invoke(e); t.append(time.perf_counter() - t0)
        res[i][0].append(np.median(t) * 1e3)
        t = []
        for _ in range(30):
            f = frame.copy(); t0 = time.perf_counter(); e.classify_frame(f, 640, 480, ya.COMPAT_SANE); t.append(time.perf_counter() - t0)
        res[i][1].append(np.median(t) * 1e3)
for i, (v, e) in enumerate(engs):
    print(f"tfl_fuse={v['tfl_fuse']} tfl_graph={v['tfl_graph']} tfl_dot={v['tfl_dot']} tfl_group={v['tfl_group']}: invoke {np.median(res[i][0]):.3f} ms (min {min(res[i][0]):.3f}), classify 640x480 {np.median(res[i][1]):.3f} ms (min {min(res[i][1]):.3f})", flush=True)
