"""Study (round 5): does the detection tail, run inside the engine's step, give the SAME bits as the oracle's tail run on the
engine's own head outputs - step after step, in every execution mode?  (Equal by construction; a difference is a race, a stale
buffer or a miscompiled kernel.)  Two 550x550 noise frames are alternated, so anything left over from the step before is wrong
data; `idle` seconds of idle device before each step.  Also checks that heads and prototypes of a frame repeat bit for bit two
steps later.  A differing detection is printed with its place in K1's grid (workgroup, lane, wave).

    python tools/study/tail_vs_oracle_repeat.py [reps = 6] [idle_s = 0] [modes = eager1,eager2,graph1,graph2t,graph2] [frames per step = 1] [r50_550_f16 | r101_700_fp8]

What it found: tools/study/k1_wide_reads_packed_exp.patch (a K1 whose LDS reads the compiler merged into wide ones) fails in mode
graph2 with idle >= 2 s in about a third of the steps (the rate moves with the box; clean runs of 12-16 steps happen), always in
lanes 48-63 of a wave; the forms with one ds_read_u16 per logit have not failed in 400 steps.
YH_STUDY_LIB=<path> runs a study build of the library instead of the in-tree one. Uses the oracle: a test tool, not product code."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd")); sys.path.insert(0, ROOT)
import yolact_amd as ya
from oracle import oracle
if os.environ.get("YH_STUDY_LIB"):   # a study build of the kernel library instead of the in-tree one (this TOOL reads the variable, the library reads none)
    from yolact_amd import capi
    capi.lib_path = lambda: os.path.abspath(os.environ["YH_STUDY_LIB"])

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # seconds of idle device before each step
CONFIG = sys.argv[5] if len(sys.argv) > 5 else "r50_550_f16"
S = 700 if CONFIG == "r101_700_fp8" else 550
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # frames per step (the tail is checked on the first four)
imgs = [np.random.default_rng(5 + k).integers(0, 256, (N, S, S, 3), dtype=np.uint8) for k in range(2)]   # alternated: data left over from the step before is WRONG data
MODES = {"eager1": ("eager, one stream", dict(use_graph=False, tune=dict(headfork_maxb=0, tailfork=0))),
         "eager2": ("eager, two streams", dict(use_graph=False)),
         "graph1": ("graph, one stream (+ the dummy branch)", dict(use_graph=True, tune=dict(headfork_maxb=0, tailfork=0))),
         "graph2t": ("graph, tail fork only", dict(use_graph=True, tune=dict(headfork_maxb=0))),
         "graph2": ("graph, two streams", dict(use_graph=True))}
for label, kw in (MODES[m] for m in (sys.argv[3].split(",") if len(sys.argv) > 3 else MODES)):
    if CONFIG == "r101_700_fp8":   # configs[4]'s engine: R101, 700 x 700, E4M3 operands on the K-heavy 3x3 layers
        kw = dict(kw, backbone=101, precision=ya.PRECISION_FP8)
    eng = ya.Engine(input_size=S, max_batch=max(8, N), **kw)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    if CONFIG == "r101_700_fp8":
        eng.set_input(imgs[0]); eng.fp8_calibrate()
    pri = eng.priors()
    prev = None; prev_scores = None; hist = [None, None]
    for r in range(reps):
        time.sleep(idle)
        eng.set_input(imgs[r & 1]); eng.evaluate()
        got = [eng.output(i) for i in range(4)]
        dets, masks = eng.detections(0)
        if r >= 2:   # the same frame two steps ago: heads and prototypes must repeat bit for bit
            for name, x, y in zip(("loc", "conf", "mask", "proto"), got, hist[r & 1]):
                nd = int(np.count_nonzero(x.view(np.uint32) != y.view(np.uint32)))
                if nd: print(f"{label} rep {r}: {name} differs from the same frame's earlier step in {nd} of {x.size} values", flush=True)
        hist[r & 1] = got
        for f in range(1, min(N, 4)):   # further frames of the step: counted only
            fd, fm = eng.detections(f)
            od, om = oracle.detect(got[0][f], got[1][f], got[2][f], got[3][f], pri)
            nb = sum(1 for x, y in zip(fd, od) if (x["class_id"], x["prior"], x["score"], x["box"]) != (y["class_id"], y["prior"], y["score"], y["box"])) + abs(len(fd) - len(od))
            if nb or not np.array_equal(fm, om): print(f"{label} rep {r}: frame {f}: {nb} detections differ, masks equal: {np.array_equal(fm, om)}", flush=True)
        odets, omasks = oracle.detect(got[0][0], got[1][0], got[2][0], got[3][0], pri)
        a = [(d["class_id"], d["prior"], d["score"]) for d in dets]
        b = [(d["class_id"], d["prior"], d["score"]) for d in odets]
        bad = [(x, y) for x, y in zip(a, b) if x != y]
        print(f"{label} rep {r}: {len(a)} / {len(b)} detections, {len(bad)} differ" + (f"  first: {bad[0]}" if bad else ""), flush=True)
        if bad and hasattr(eng.L, "yh_debug_study_read"):   # study build: the raw dwords every lane read from LDS, against the head rows
            import ctypes
            raw = np.zeros((128, 192, 41), np.uint32)
            eng.L.yh_debug_study_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
            assert eng.L.yh_debug_study_read(raw.ctypes.data, raw.nbytes) == 0
            conf = got[1][0].astype(np.float16)                     # [P, 81], exact (the heads are f16)
            prev_conf = prev[1][0].astype(np.float16) if prev is not None else None
            for x, y in bad:
                pr = x[1]; wg, lane = pr // 192, pr % 192
                want = np.zeros(82, np.float16); want[:81] = conf[pr]
                words = raw[wg, lane]
                gotv = np.zeros(82, np.float16)
                gotv[0] = np.array([words[0] & 0xFFFF], np.uint16).view(np.float16)[0]
                gotv[1:81] = words[1:].view(np.float16)[:80]
                diff = np.nonzero(gotv[:81].view(np.uint16) != want[:81].view(np.uint16))[0]
                desc = []
                for c in diff:
                    stale = prev_conf is not None and gotv[c].view(np.uint16) == prev_conf[pr][c].view(np.uint16)
                    desc.append(f"c{c}: read {float(gotv[c]):.4g} want {float(want[c]):.4g}" + (" (= previous step's)" if stale else ""))
                print(f"      prior {pr} workgroup {wg} lane {lane}: {len(diff)} of 81 logits read wrong: " + "; ".join(desc[:8]))
        if bad and prev_scores is not None:   # is the wrong score the PREVIOUS step's score of the same (class, prior)?  (a candidate entry left over)
            for x, y in bad:
                ps = prev_scores.get((x[0], x[1]))
                print(f"      class {x[0]} prior {x[1]}: engine {x[2]!r}, this step's oracle {y[2]!r}, previous step's candidate score {ps!r}" + ("  <== EQUAL: a stale entry" if ps == x[2] else ""))
        prev = got
        prev_scores = {(d["class_id"], d["prior"]): d["score"] for d in odets}
        for x, y in bad:   # where in K1's grid: 64 cells (192 priors) per workgroup, lane = prior % 192
            print(f"      class {x[0]:2d} prior {x[1]:5d} = workgroup {x[1] // 192:3d} lane {x[1] % 192:3d} (wave {x[1] % 192 // 64})  score {x[2]:.9g} vs {y[2]:.9g}  rel {x[2] / y[2] - 1:+.2e}")
    eng.close()
