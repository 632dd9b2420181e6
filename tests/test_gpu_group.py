"""-m gpu: frame sharding as a library feature (yh_group_*: SURVEY.md §7.1 step 7, §8e) and the copy / compute overlap of
yh_set_input_* (two input buffers + a copy stream).

The reference's caller is one process that owns the frame loop (/root/reference/src/main.rs:63-75, src/scene.rs:77-92); a
group is what it calls with N frames to get N results over the GPUs of a node. On the one-GPU test box the members share
device 0 (weights then travel by device-to-device copy instead of the RCCL broadcast, which needs distinct devices: that
path is unverified on hardware until an 8-GPU node runs it). A frame's result depends only on its member's batch size, so
every member must reproduce, bit for bit, what a single engine gives for the same block of frames."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tiny-object-detection_amd", "lib", "yolact_demo")
TH = 0.005


def _single(ya, blob, S, max_batch, blocks, **kw):
    """Reference results: ONE engine, one block of frames at a time, serially."""
    e = ya.Engine(input_size=S, max_batch=max_batch, conf_thresh=TH, **dict(dict(use_graph=True), **kw))
    e.load_weights(blob)
    out = []
    for fr in blocks:
        e.set_input(fr); e.evaluate()
        out += [e.detections(f) for f in range(fr.shape[0])]
    e.close()
    return out


class _Hip:
    """Device / pinned buffers straight from the HIP runtime the library itself is linked to (no torch: its bundled HIP runtime
    cannot initialise the device once the system one has - 'no ROCm-capable device' - in a process that loaded our library first)."""

    def __init__(self):
        import ctypes as C
        import yolact_amd as ya
        ya.load_library()
        # the HIP runtime THIS process has already mapped (the one libyolact_hip.so bound to: /opt/rocm's, or torch's bundled copy
        # when some earlier test imported torch first - opening the other one would clash with the loaded ROCr)
        path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln), "libamdhip64.so")
        self.C, self.L = C, C.CDLL(path)
        self.bufs, self.pinned = [], []

    def device(self, arr):
        C = self.C
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), C.c_size_t(arr.nbytes)) == 0
        assert self.L.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0   # hipMemcpyHostToDevice
        self.bufs.append(p)
        return p.value

    def pinned_like(self, shape):
        C = self.C
        n = int(np.prod(shape))
        p = C.c_void_p()
        assert self.L.hipHostMalloc(C.byref(p), C.c_size_t(n), 0) == 0
        self.pinned.append(p)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).reshape(shape)

    def close(self):
        for p in self.bufs:
            self.L.hipFree(p)
        for p in self.pinned:
            self.L.hipHostFree(p)


def _same(a, b):
    (da, ma), (db, mb) = a, b
    return da == db and np.array_equal(ma, mb)


@pytest.mark.parametrize("S,per", [(256, 4), (550, 4)])
def test_group_of_two_members_on_one_device_equals_single_engine(built, S, per):
    import yolact_amd as ya
    g = ya.Group([0, 0], input_size=S, max_batch=per, conf_thresh=TH)
    blob = g.members[0].generate_weights(seed=1)
    g.load_weights(blob)
    assert "device-to-device" in g.weights_replication(), g.weights_replication()
    rng = np.random.default_rng(31)
    frames = rng.integers(0, 256, (2 * per, S, S, 3), dtype=np.uint8)
    # full blocks: member 0 <- frames[:per], member 1 <- frames[per:]
    g.evaluate(frames)
    got = [g.detections(f) for f in range(2 * per)]
    want = _single(ya, blob, S, per, [frames[:per], frames[per:]])
    assert sum(len(d) for d, _ in want) >= 2 * per
    assert [g.frame_owner(f) for f in (0, per - 1, per, 2 * per - 1)] == [(0, 0), (0, per - 1), (1, 0), (1, per - 1)]
    for f in range(2 * per):
        assert _same(got[f], want[f]), f
    # ragged: 2 * per - 1 frames -> blocks of per and per - 1; and a call that leaves member 1 idle
    g.evaluate(frames[: 2 * per - 1])
    want = _single(ya, blob, S, per, [frames[:per], frames[per: 2 * per - 1]])
    for f in range(2 * per - 1):
        assert _same(g.detections(f), want[f]), f
    g.evaluate(frames[:1])
    assert g.frame_owner(0) == (0, 0) and _same(g.detections(0), _single(ya, blob, S, per, [frames[:1]])[0])
    with pytest.raises(ya.YhError):
        g.detections(1)                                   # out of the last call's range
    with pytest.raises(ya.YhError):
        g.evaluate(np.zeros((2 * per + 1, S, S, 3), np.uint8))   # more than members * max_batch
    # back-to-back calls pipeline (no sync in between): the last call's results are what is read
    for k in range(3):
        g.evaluate(np.roll(frames, k, axis=0))
    want = _single(ya, blob, S, per, [np.roll(frames, 2, axis=0)[:per], np.roll(frames, 2, axis=0)[per:]])
    for f in range(2 * per):
        assert _same(g.detections(f), want[f]), f
    g.close()


def test_group_prepares_on_the_callers_thread_and_workers_never_capture(built):
    """The setup discipline of DESIGN.md section 7: every graph capture of a group happens on the CALLING thread, one member after the
    other, while no worker thread is inside a HIP call. Two block sizes alternate (5 frames -> blocks 3 + 2, 6 -> 3 + 3, 2 -> 1 + 1:
    member 1 meets three shapes, two of them only after the group has been running), a third appears late, and the process-wide
    audit counters must show captures and worker jobs but NOT ONE overlap of the two; results stay bit-equal to a single engine.
    (Round 4: a member capturing beside a neighbour's once-only allocation, both on worker threads, ended a full test run with a
    host segfault. The group's first steps on threads - capture + first host input - are exactly what this test takes, once.)"""
    import yolact_amd as ya
    S, per = 256, 3
    a0 = ya.setup_audit()
    g = ya.Group([0, 0], input_size=S, max_batch=per, conf_thresh=TH)
    blob = g.members[0].generate_weights(seed=1)
    g.load_weights(blob)
    g.prepare(5)                                           # explicit: blocks of 3 and 2, both input buffers each = 4 captures
    a1 = ya.setup_audit()
    assert a1["setups"] - a0["setups"] == 4 and a1["worker_jobs"] == a0["worker_jobs"], (a0, a1)
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (6, S, S, 3), dtype=np.uint8)
    for it, n in enumerate((5, 6, 5, 6, 5, 2, 6, 5)):      # 6 -> member 1 needs a block of 3 (late capture), 2 -> both need blocks of 1
        g.evaluate(frames[:n])
        want = _single(ya, blob, S, per, [frames[:(n + 1) // 2], frames[(n + 1) // 2:n]])
        for f in range(n):
            assert _same(g.detections(f), want[f]), (it, n, f)
    a2 = ya.setup_audit()
    # captures after prepare: member 1 at 3 frames (2) + both members at 1 frame (4); _single's engines capture too (on this thread)
    assert a2["setups"] - a1["setups"] >= 6 and a2["worker_jobs"] - a1["worker_jobs"] == 16, (a1, a2)
    assert a2["overlaps"] == a0["overlaps"] == 0 and a2["in_flight"] == 0, (a0, a2)
    # a member handed to a worker with an unprepared shape refuses instead of capturing there: forced through the handle's own
    # entry points this cannot be reached from outside, so the rule is checked where it lives - eager members have nothing to
    # capture and take their first step on the caller's thread
    ge = ya.Group([0, 0], input_size=S, max_batch=per, conf_thresh=TH, use_graph=False)
    ge.load_weights(blob)
    b0 = ya.setup_audit()
    ge.evaluate(frames[:5])                                # first step of both members: inline, no worker job
    b1 = ya.setup_audit()
    assert b1["worker_jobs"] == b0["worker_jobs"] and b1["setups"] == b0["setups"], (b0, b1)
    ge.evaluate(frames[:5])                                # second: on the workers
    b2 = ya.setup_audit()
    assert b2["worker_jobs"] - b1["worker_jobs"] == 2 and b2["overlaps"] == 0, (b1, b2)
    want = _single(ya, blob, S, per, [frames[:3], frames[3:5]], use_graph=False)
    for f in range(5):
        assert _same(ge.detections(f), want[f]), f
    g.close(); ge.close()


def test_group_evaluate_device_with_resident_frames(built):
    import yolact_amd as ya
    S, per = 256, 3
    g = ya.Group([0, 0, 0], input_size=S, max_batch=per, conf_thresh=TH)
    blob = g.members[0].generate_weights(seed=1)
    g.load_weights(blob)
    hip = _Hip()
    host = [np.random.default_rng(70 + i).integers(0, 256, (per, S, S, 3), dtype=np.uint8) for i in range(3)]
    ptrs = [hip.device(a) for a in host]
    g.evaluate_device(ptrs, [3, 0, 2])                                   # member 1 sits out
    got = [g.detections(f) for f in range(5)]
    want = _single(ya, blob, S, per, [host[0], host[2][:2]])
    assert g.frame_owner(3) == (2, 0)
    for f in range(5):
        assert _same(got[f], want[f]), f
    g.close(); hip.close()


def test_cpp_group_host_matches_python(built, tmp_path):
    import yolact_amd as ya
    S, per, n = 256, 3, 5
    frames = np.random.default_rng(3).integers(0, 256, (n, S, S, 3), dtype=np.uint8)
    fin, fout = tmp_path / "frames.u8", tmp_path / "dets.txt"
    frames.tofile(fin)
    r = subprocess.run([DEMO, "--group", str(fin), str(fout), str(n), str(S), str(per), "0,0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "2 members" in r.stdout and "device-to-device" in r.stdout
    g = ya.Group([0, 0], input_size=S, max_batch=per)                    # default threshold, as the C++ host
    blob = g.members[0].generate_weights(seed=1)
    g.load_weights(blob)
    g.evaluate(frames)
    lines = []
    for f in range(n):
        for d in g.detections(f, want_masks=False)[0]:
            bits = np.array([d["score"], *d["box"]], np.float32).view(np.uint32)
            lines.append("%d %d %d %08x %08x %08x %08x %08x" % (f, d["class_id"], d["prior"], *bits))
    g.close()
    assert fout.read_text().split("\n")[:-1] == lines


@pytest.mark.parametrize("n", [1, 8])
def test_set_input_overlaps_the_running_step_and_changes_no_bit(built, n):
    """yh_set_input_u8 of frame batch k+1 is issued while step k runs (its copy goes to the other input buffer on the copy
    stream), the host buffer is overwritten as soon as the call returns (copy_from_slice semantics, src/yolact.rs:161-162),
    and every step's heads, detections and masks equal the serial order bit for bit. n = 1: pinned staging path (<= 4 MB);
    n = 8: the runtime's pageable path; plus a pinned (torch) source for both."""
    import yolact_amd as ya
    S, K = 550, 7
    rng = np.random.default_rng(41 + n)
    F = rng.integers(0, 256, (K, n, S, S, 3), dtype=np.uint8)
    eng = ya.Engine(input_size=S, max_batch=n, use_graph=True)
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    want = []
    for k in range(K):                                   # serial: copy, step, read
        eng.set_input(F[k]); eng.evaluate()
        want.append(([eng.output(i) for i in (1, 3)], [eng.detections(f) for f in range(n)]))

    def read():
        return [eng.output(i) for i in (1, 3)], [eng.detections(f) for f in range(n)]

    def check(got):
        for k in range(K):
            for a, b in zip(got[k][0], want[k][0]):
                assert np.array_equal(a, b), k
            for a, b in zip(got[k][1], want[k][1]):
                assert _same(a, b), k
    hip = _Hip()
    sources = [np.empty((n, S, S, 3), np.uint8), hip.pinned_like((n, S, S, 3))]     # pageable, pinned (hipHostMalloc)
    for buf in sources:
        got = []
        buf[:] = F[0]; eng.set_input(buf); buf[:] = 0xAB
        eng.evaluate()
        for k in range(1, K):
            buf[:] = F[k]; eng.set_input(buf); buf[:] = 0xAB      # copy k runs under step k - 1; the source is reused at once
            got.append(read())                                     # results of step k - 1 (the main stream is waited for)
            eng.evaluate()
        got.append(read())
        check(got)
    # steps that re-use the frames last set (no yh_set_input_* in between) read the same buffer again
    eng.evaluate()
    assert all(np.array_equal(a, b) for a, b in zip(read()[0], want[K - 1][0]))
    eng.close(); hip.close()


def test_bench_single_process_mode_prints_the_contract_line(built):
    """bench.py --single-process: the N-GPU path driven by ONE process through yh_group_* (two members sharing device 0 here)."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--steps", "3", "--warmup", "2", "--batch", "4"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["global_batch"] == 8 and b["scaling"] == "weak" and "yh_group" in b["config"]["workload"]
    assert abs(b["value"] - 2 * 4 * 3 / (b["ms_per_step"] * 3e-3)) < 0.01 * b["value"] and b["roofline"]["frac"] > 0
