#!/usr/bin/env python3
"""Where does a host-fed step spend its time? Host seconds inside yh_set_input_u8 / yh_evaluate and the resulting rate, per batch size,
pageable and pinned sources (bench.py's pcie_inclusive_fps is the pageable case)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya  # noqa: E402

ya.load_library()
hip = C.CDLL(next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln), "libamdhip64.so"))   # the runtime already mapped
S = 550
src = ya.Engine(input_size=S, max_batch=1, use_graph=False)
blob = src.generate_weights(1)
for n in (1, 8, 64):
    eng = ya.Engine(input_size=S, max_batch=n, use_graph=True)
    eng.load_weights(blob)
    pageable = np.random.default_rng(0).integers(0, 256, (n, S, S, 3), dtype=np.uint8)
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), C.c_size_t(pageable.nbytes), 0) == 0
    pinned = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(pageable.size,)).reshape(pageable.shape)
    pinned[:] = pageable
    eng.set_input(pageable); eng.evaluate(); eng.sync()
    t_res = eng.time_steps(20) / 20
    for tag, host in (("pageable", pageable), ("pinned", pinned)):
        steps = 200 if n == 1 else (60 if n == 8 else 20)
        eng.set_input(host); eng.evaluate(); eng.sync()
        ts = te = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            a = time.perf_counter(); eng.set_input(host); b = time.perf_counter(); eng.evaluate(); c = time.perf_counter()
            ts += b - a; te += c - b
        eng.sync()
        dt = time.perf_counter() - t0
        print(f"batch {n:2d} {tag:8s}: {n * steps / dt:8.1f} frames/s host-fed ({dt / steps * 1e3:.3f} ms per step; resident step {t_res:.3f} ms) "
              f"- host time in yh_set_input_u8 {ts / steps * 1e3:.3f} ms, in yh_evaluate {te / steps * 1e3:.3f} ms")
    hip.hipHostFree(p)
    eng.close()
