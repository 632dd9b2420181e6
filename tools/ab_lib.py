"""A/B of two BUILDS of the kernel library on one box: graph-replayed step time, the two libraries alternated process by process.
    python tools/ab_lib.py <libA.so> <libB.so> [batch = 64] [rounds = 4]
(build the other one with `make -C tiny-object-detection_amd BUILD=build_old LIBDIR=lib_old` in a checkout of the older tree)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(lib, batch):
    sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
    import yolact_amd as ya
    from yolact_amd import capi
    capi.lib_path = lambda: lib
    eng = ya.Engine(input_size=550, max_batch=batch, use_graph=True)
    eng.load_weights(eng.generate_weights(1))
    eng.set_input(np.random.default_rng(0).integers(0, 256, (batch, 550, 550, 3), dtype=np.uint8))
    for _ in range(3):
        eng.evaluate()
    eng.sync()
    steps = 100 if batch <= 8 else 20
    best = min(eng.time_steps(steps, True) / steps for _ in range(3))
    print(f"{best:.5f}")

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
        batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
        rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
        ms = [[], []]
        for r in range(rounds):
            for i in (0, 1):
                out = subprocess.run([sys.executable, __file__, "--child", libs[i], str(batch)], capture_output=True, text=True, check=True).stdout
                ms[i].append(float(out.strip().splitlines()[-1]))
                print(f"round {r} {'AB'[i]} {os.path.relpath(libs[i], ROOT)}: {ms[i][-1]:.4f} ms/step", flush=True)
        for i in (0, 1):
            print(f"{'AB'[i]}: median {np.median(ms[i]):.4f} ms, min {min(ms[i]):.4f}  (batch {batch}, best of 3 x timed replays per process)")
