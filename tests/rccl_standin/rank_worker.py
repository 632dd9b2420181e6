"""One rank of the two-process weight-broadcast test (tests/test_gpu_rccl_standin.py). TEST INFRASTRUCTURE: run as a subprocess whose
LD_LIBRARY_PATH puts the stand-in librccl.so.1 (librccl_standin.c, built by the test) in front, so that the library's unchanged
dlopen("librccl.so.1") finds it. No torch here: a process that has torch loaded already holds torch's bundled librccl under that soname.

usage: rank_worker.py RANK NRANKS ID_FILE OUT_FILE [SIZE]
Rank 0 generates and loads the weights, draws the unique id through the library (yh_rccl_unique_id) and leaves it in ID_FILE; every
rank calls yh_rank_broadcast_weights, then runs two seeded frames and writes what the test compares: the sha256 of the canonical blob
as it sits in device memory, of the four head outputs, and the detections. A library error is reported (with yh_last_error's text) as
"ERROR <code> <message>"; the handle must stay usable after it, which the worker checks by loading weights through the host and
running the same frames ("AFTER_ERROR ok <sha of the heads>")."""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))
import yolact_amd as ya   # noqa: E402


def device_bytes(ptr, n):
    """The handle's canonical blob, read back through the HIP runtime the library itself is bound to."""
    path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln), "libamdhip64.so")
    hip = C.CDLL(path)
    buf = np.empty(n, np.uint8)
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(n), 2) == 0   # hipMemcpyDeviceToHost
    return buf


def run(eng, S):
    frames = np.random.default_rng(77).integers(0, 256, (2, S, S, 3), dtype=np.uint8)
    eng.set_input(frames)
    eng.evaluate()
    heads = hashlib.sha256(b"".join(np.ascontiguousarray(eng.output(i)).tobytes() for i in range(4))).hexdigest()
    dets = []
    for f in range(2):
        d, m = eng.detections(f)
        dets.append((repr(d), hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest()))
    return heads, dets


def main():
    rank, nranks, id_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    S = int(sys.argv[5]) if len(sys.argv) > 5 else 160
    lines = []
    eng = ya.Engine(input_size=S, max_batch=2, use_graph=True, conf_thresh=0.005)
    blob = eng.generate_weights(seed=1)
    try:
        if rank == 0:
            eng.load_weights(blob)
            ident = ya.rccl_unique_id()
            with open(id_file + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(id_file + ".tmp", id_file)
        else:
            t0 = time.time()
            while not os.path.exists(id_file):
                if time.time() - t0 > 60:
                    raise RuntimeError("no id file")
                time.sleep(0.02)
            ident = open(id_file, "rb").read()
        assert len(ident) == 128
        eng.rank_broadcast_weights(ident, rank, nranks, 0)
        n = eng.weights_nbytes()
        lines.append("BLOB " + hashlib.sha256(device_bytes(eng.weights_device_ptr(), n).tobytes()).hexdigest())
        lines.append("WANT " + hashlib.sha256(np.ascontiguousarray(blob).tobytes()).hexdigest())
        heads, dets = run(eng, S)
        lines.append("HEADS " + heads)
        lines += [f"DETS {f} {d} {m}" for f, (d, m) in enumerate(dets)]
    except ya.YhError as e:
        lines.append(f"ERROR {e.code} {e}")
        eng.load_weights(blob)                 # the handle is still usable: host load + the same frames
        heads, _ = run(eng, S)
        lines.append("AFTER_ERROR ok " + heads)
    eng.close()
    with open(out_file + ".tmp", "w") as f:
        f.write("\n".join(lines) + "\n")
    os.replace(out_file + ".tmp", out_file)


if __name__ == "__main__":
    main()
