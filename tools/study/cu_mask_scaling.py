"""VERDICT r4 item 4 - the one bounded experiment on overlapping the step's two bounds.

The batch-64 step is 5.4 ms of backbone (bound by each CU's own vector-memory path: HBM-side 3-4 TB/s, MFMA pipes mostly idle) followed
by 4 ms of protonet + head (MFMA-bound at the rate the chip sustains, HBM idle). Two engines staggered by half a step on DISJOINT CU
sets (hipExtStreamCreateWithCUMask) would keep both resources busy all the time - IF a phase loses less than proportionally when it
is given fewer CUs (memory-bound kernels because HBM, not the CU, was their bound; matrix-bound kernels because fewer MFMA CUs hold
a higher clock under the power cap). This script measures exactly that, per kernel family, with eager launches (a CU mask belongs
to a stream; a captured graph's side branches run on the runtime's own streams):

  part 1  one engine, batch 64, per-launch hipEvent times (yh_profile_run) with its streams on 256 / 192 / 160 / 128 / 96 CUs:
          t(k CUs) / t(256) per family against the proportional 256 / k;
  part 2  two engines, batch 32 each, on disjoint sets (A: k CUs, B: 256 - k), both looping eagerly from two host threads with
          the second started half a step late: frames/s of the pair against ONE engine at batch 64 on the whole chip (graph replay,
          the shipped form) and against one eager engine at batch 64.
Kill criterion (written before the run): if the best pair is < 3 % over the single engine, the item is closed."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
sys.path.insert(0, ROOT)
import yolact_amd as ya
import bench

S = 550
rng = np.random.default_rng(0)


def fam_times(prof):
    by = {}
    for p in prof:
        k = bench.kernel_family(p["name"].split(":")[0])
        by[k] = by.get(k, 0.0) + p["ms"]
    return by


def part1():
    n = 64
    eng = ya.Engine(input_size=S, max_batch=n, use_graph=False)
    eng.load_weights(eng.generate_weights(1))
    eng.set_input(rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8))
    eng.evaluate(); eng.sync()
    res = {}
    for k in (256, 192, 160, 128, 96, 256):
        eng.set_cu_mask(k)
        eng.evaluate(); eng.sync()
        prof = eng.profile(True, 3)
        res.setdefault(k, []).append((sum(p["ms"] for p in prof), fam_times(prof), {p["name"]: p["ms"] for p in prof}))
    base_total, base_f, base_l = res[256][0]
    print(f"part 1: one engine, batch {n}, eager, per-launch hipEvent sums (serialised). 256 CUs measured twice: {res[256][0][0]:.3f} / {res[256][1][0]:.3f} ms")
    fams = sorted(base_f, key=lambda f: -base_f[f])[:12]
    print(f"{'family':44s} {'ms@256':>8s} " + " ".join(f"{'x@' + str(k):>8s}" for k in (192, 160, 128, 96)) + "   (proportional: 1.33 1.60 2.00 2.67)")
    for f in fams:
        print(f"{f:44s} {base_f[f]:8.3f} " + " ".join(f"{res[k][0][1].get(f, 0) / base_f[f]:8.2f}" for k in (192, 160, 128, 96)))
    print(f"{'whole step':44s} {base_total:8.3f} " + " ".join(f"{res[k][0][0] / base_total:8.2f}" for k in (192, 160, 128, 96)))
    # the two phases: launches up to and including the FPN laterals / everything after (protonet, head, tail)
    names = list(base_l)
    cut = max(i for i, nm in enumerate(names) if ":lat" in nm or ":c5" in nm or ":l4b" in nm) + 1
    phase = {}
    for label, sel in (("backbone + laterals", names[:cut]), ("protonet + head + tail", names[cut:])):
        b = sum(base_l[nm] for nm in sel)
        phase[label] = {k: sum(res[k][0][2][nm] for nm in sel) for k in (256, 192, 160, 128, 96)}
        print(f"{label:44s} {b:8.3f} " + " ".join(f"{phase[label][k] / b:8.2f}" for k in (192, 160, 128, 96)))
    # what the per-phase arrangement of the review would give: the chip split into a backbone region of x CUs and a matrix region of
    # 256 - x, two engines alternating between them half a step apart: one 64-frame step leaves every max(T_bb(x), T_mm(256 - x))
    print("predicted per-phase pipeline (backbone region x CUs | matrix region 256 - x): ms per 64 frames vs the serial step")
    for x in (96, 128, 160):
        tb, tm = phase["backbone + laterals"][x], phase["protonet + head + tail"][256 - x]
        print(f"  x = {x:3d}: max({tb:.3f}, {tm:.3f}) = {max(tb, tm):.3f} ms   (serial, whole chip: {base_total:.3f} ms)")
    eng.close()


def loop(eng, frames_dev_setter, steps, out, delay):
    time.sleep(delay)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.evaluate()
    eng.sync()
    out.append((t0, time.perf_counter()))


def part2():
    print("part 2: two engines at batch 32 on disjoint CU sets, eager, two host threads, B started half a step late")
    one = ya.Engine(input_size=S, max_batch=64, use_graph=True)
    blob = one.generate_weights(1)
    one.load_weights(blob)
    one.set_input(rng.integers(0, 256, (64, S, S, 3), dtype=np.uint8))
    for _ in range(3): one.evaluate()
    one.sync()
    t0 = time.perf_counter()
    for _ in range(20): one.evaluate()
    one.sync()
    ref = 64 * 20 / (time.perf_counter() - t0)
    print(f"  one engine, batch 64, graph replay, whole chip: {ref:8.1f} frames/s ({64e3 / ref:.3f} ms per 64 frames)")
    one.close()
    one = ya.Engine(input_size=S, max_batch=64, use_graph=False)
    one.load_weights(blob)
    one.set_input(rng.integers(0, 256, (64, S, S, 3), dtype=np.uint8))
    for _ in range(3): one.evaluate()
    one.sync()
    t0 = time.perf_counter()
    for _ in range(20): one.evaluate()
    one.sync()
    ref_e = 64 * 20 / (time.perf_counter() - t0)
    print(f"  one engine, batch 64, eager, whole chip:        {ref_e:8.1f} frames/s")
    one.close()
    engs = []
    for _ in range(2):
        e = ya.Engine(input_size=S, max_batch=32, use_graph=False)
        e.load_weights(blob)
        e.set_input(rng.integers(0, 256, (32, S, S, 3), dtype=np.uint8))
        e.evaluate(); e.sync()
        engs.append(e)
    half = 32e3 / ref / 1e3 * 1.0   # seconds: about half of a 64-frame step
    for ka in (None, 128, 96, 160, 112, 144):
        if ka is None:
            for e in engs: e.set_cu_mask(256)
            label = "both on all 256 CUs (no masks)"
        else:
            engs[0].set_cu_mask(ka, offset=0); engs[1].set_cu_mask(256 - ka, offset=ka)
            label = f"A on {ka} CUs, B on {256 - ka}"
        for e in engs: e.evaluate(); e.sync()
        best = 0.0
        for rep in range(3):
            outs = [[], []]
            th = [threading.Thread(target=loop, args=(engs[i], None, 24, outs[i], i * half)) for i in range(2)]
            for t in th: t.start()
            for t in th: t.join()
            span = max(o[0][1] for o in outs) - min(o[0][0] for o in outs)
            best = max(best, 2 * 32 * 24 / span)
        print(f"  {label:34s}: {best:8.1f} frames/s = {best / ref:.3f} x the single engine")
    for e in engs: e.close()


def part3():
    """The per-phase arrangement itself: engine A runs ONLY the memory-bound phase (backbone + laterals) on a region of x CUs, engine B
    ONLY the matrix-bound phase (FPN 3x3, protonet, head) on the other 256 - x, each from its own host thread, batch 64 each: time per
    phase alone on its region, and with the other region busy. Whatever one 64-frame step costs the slower region is the pipeline's
    period (two engines alternating between the regions half a step apart)."""
    print("part 3: phase 0 (backbone + laterals) on x CUs | phase 1 (FPN 3x3, protonet, head) on 256 - x CUs, batch 64, eager")
    engs = []
    for _ in range(2):
        e = ya.Engine(input_size=S, max_batch=64, use_graph=False)
        if not engs:
            blob = e.generate_weights(1)
        e.load_weights(blob)
        e.set_input(rng.integers(0, 256, (64, S, S, 3), dtype=np.uint8))
        e.evaluate(); e.sync()
        engs.append(e)
    a, b = engs
    t0, t1 = a.run_phase(0, 3) / 3, b.run_phase(1, 3) / 3
    print(f"  whole chip, alone: phase 0 {t0:.3f} ms, phase 1 {t1:.3f} ms, sum {t0 + t1:.3f} ms per 64 frames")
    for x in (96, 112, 128, 144, 160):
        a.set_cu_mask(x, offset=0); b.set_cu_mask(256 - x, offset=x)
        a.run_phase(0, 1); b.run_phase(1, 1)
        alone0, alone1 = a.run_phase(0, 4) / 4, b.run_phase(1, 4) / 4
        out = {}

        def go(e, ph, key, reps):
            out[key] = e.run_phase(ph, reps) / reps
        # both busy for the same wall time: reps in proportion to the phases' own durations
        r0, r1 = max(2, round(60 / alone0)), max(2, round(60 / alone1))
        th = [threading.Thread(target=go, args=(a, 0, 0, r0)), threading.Thread(target=go, args=(b, 1, 1, r1))]
        for t in th: t.start()
        for t in th: t.join()
        period = max(out[0], out[1])
        print(f"  x = {x:3d}: alone {alone0:.3f} | {alone1:.3f} ms; together {out[0]:.3f} | {out[1]:.3f} ms -> period {period:.3f} ms per 64 frames = "
              f"{(t0 + t1) / period:.3f} x the serial phases")
    for e in engs: e.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "3"]
    if "1" in which: part1()
    if "2" in which: part2()
    if "3" in which: part3()
