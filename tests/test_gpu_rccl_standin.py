"""-m gpu: the path's one collective with MORE THAN ONE RANK, executed on one GPU behind a stand-in librccl (SURVEY.md section 8e;
BASELINE.json configs[3] / [4] start with it; the reference has no counterpart - one process, one device, src/main.rs:63-75).

Real RCCL refuses two ranks on one device and the test box has one device, so yh_rank_broadcast_weights / yh_group_broadcast_weights
(csrc/engine.hip) had only ever executed with n = 1: the by-value ncclUniqueId, ncclUint8 = 1, the stream order of the broadcast and
the communicator's lifetime are marshalled by hand against a dlopen'ed library. tests/rccl_standin/librccl_standin.c implements the
eight entry points the library binds, with the real signatures, over shared memory + hipMemcpyAsync on the caller's stream; it is
built HERE into a temp directory and found by the loader only in the subprocesses this test starts (LD_LIBRARY_PATH; the library's
dlopen("librccl.so.1") is unchanged and nothing of this ships). What this proves: the library's side of the protocol. What it cannot:
RCCL's own transport over xGMI - that waits for an 8-GPU node."""
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "rccl_standin", "librccl_standin.c")
WORKER = os.path.join(ROOT, "tests", "rccl_standin", "rank_worker.py")


@pytest.fixture(scope="module")
def standin(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("rccl_standin")
    so = d / "librccl.so.1"
    subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-I/opt/rocm/include", "-o", str(so), SRC, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    yield str(d), str(so)
    for f in glob.glob("/dev/shm/rccl_standin_*"):   # (segments a failed rendezvous left behind)
        os.unlink(f)


def _env(libdir, **extra):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update({k: str(v) for k, v in extra.items()})
    return env


def _ranks(tmp_path, libdir, n=2, envs=None, only=None, timeout=300):
    """Starts ranks 0..n-1 (or `only`) as processes on device 0; returns {rank: lines of its output file}."""
    idf = str(tmp_path / "id.bin")
    procs = {}
    for r in (only if only is not None else range(n)):
        out = str(tmp_path / f"rank{r}.txt")
        procs[r] = (subprocess.Popen([sys.executable, WORKER, str(r), str(n), idf, out], env=_env(libdir, **((envs or {}).get(r, {}))),
                                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), out)
    res = {}
    for r, (p, out) in procs.items():
        log, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, f"rank {r}:\n{log[-3000:]}"
        res[r] = open(out).read().splitlines()
    return res


def test_two_processes_broadcast_on_one_device(standin, tmp_path):
    """ncclGetUniqueId -> (file) -> ncclCommInitRank x 2 -> ncclBroadcast -> ncclCommDestroy, two processes on device 0: rank 1's
    canonical blob in device memory is byte-equal to rank 0's and to the generated one, its panels give bit-equal heads and
    detections (rank 1 never saw the weights any other way)."""
    libdir, _ = standin
    res = _ranks(tmp_path, libdir)
    r0, r1 = res[0], res[1]
    assert not any(ln.startswith("ERROR") for ln in r0 + r1), (r0, r1)
    blob0, want0 = r0[0].split()[1], r0[1].split()[1]
    assert blob0 == want0 and r1[0].split()[1] == blob0, (r0[:2], r1[:2])
    assert r0[2].startswith("HEADS ") and r0[2:] == r1[2:] and len(r0) == 5, (r0, r1)
    assert "[]" not in r0[3], r0[3]   # (the comparison is of real detections)


@pytest.mark.parametrize("fail,where", [("initrank", "ncclCommInitRank"), ("broadcast", "ncclBroadcast")])
def test_collective_errors_come_back_as_yh_ehip_and_leave_the_handle_usable(standin, tmp_path, fail, where):
    """Rank 1's ncclCommInitRank / ncclBroadcast fails (fault injection in the stand-in): rank 1 gets YH_EHIP naming the call and
    RCCL's error string; rank 0, whose peer never arrives / never takes the data, is released by the stand-in's deadline (real RCCL
    would block: the hang mode INTEGRATION.md section 4 tells hosts to guard) and gets YH_EHIP too; both handles then load weights
    through the host and run - same heads on both."""
    libdir, _ = standin
    res = _ranks(tmp_path, libdir, envs={0: dict(RCCL_STANDIN_TIMEOUT_S=4), 1: dict(RCCL_STANDIN_FAIL=fail, RCCL_STANDIN_TIMEOUT_S=4)})
    for r in (0, 1):
        assert res[r][0].startswith("ERROR -2 "), res[r]
        assert "stand-in librccl" in res[r][0], res[r]
        assert res[r][1].startswith("AFTER_ERROR ok "), res[r]
    assert where in res[1][0], res[1]
    assert res[0][1] == res[1][1]


def test_a_peer_that_never_arrives(standin, tmp_path):
    """Rank 0 alone with nranks = 2: the rendezvous gives up at the stand-in's deadline, the library reports YH_EHIP from
    ncclCommInitRank and the handle works afterwards."""
    libdir, _ = standin
    res = _ranks(tmp_path, libdir, only=[0], envs={0: dict(RCCL_STANDIN_TIMEOUT_S=3)})
    assert res[0][0].startswith("ERROR -2 ") and "ncclCommInitRank" in res[0][0] and res[0][1].startswith("AFTER_ERROR ok "), res[0]


GROUP_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import yolact_amd as ya
L = ya.load_library()
assert L.yh_debug_rccl_shared_device(1) == 0
S, per = 160, 2
g = ya.Group([0, 0, 0], input_size=S, max_batch=per, conf_thresh=0.005)
blob = g.members[0].generate_weights(seed=1)
mode = sys.argv[2]
if mode == "fail":
    try:
        g.load_weights(blob)
        print("HOW", g.weights_replication())
    except ya.YhError as e:
        print("ERROR", e.code, e)
else:
    g.load_weights(blob)
    print("HOW", g.weights_replication())
frames = np.random.default_rng(3).integers(0, 256, (3 * per, S, S, 3), dtype=np.uint8)
g.evaluate(frames)
one = ya.Engine(input_size=S, max_batch=per, use_graph=True, conf_thresh=0.005)
one.load_weights(blob)
ok = True
for b in range(3):
    one.set_input(frames[b * per:(b + 1) * per]); one.evaluate()
    for f in range(per):
        (da, ma), (db, mb) = g.detections(b * per + f), one.detections(f)
        ok = ok and da == db and np.array_equal(ma, mb) and len(da) > 0
print("EQUAL", ok)
g.close(); one.close()
"""


def test_group_replication_takes_its_rccl_branch(standin, tmp_path):
    """yh_group_replicate_weights with its RCCL branch forced (yh_debug_rccl_shared_device: three members on device 0 count as three
    ranks): ncclCommInitAll + ncclGroupStart / three ncclBroadcast / ncclGroupEnd on the members' own streams, then every member's
    detections equal a single engine's. With ncclCommInitAll failing the group falls back to the host path and says why."""
    libdir, _ = standin
    pkg = os.path.join(ROOT, "tiny-object-detection_amd")
    r = subprocess.run([sys.executable, "-c", GROUP_SCRIPT, pkg, "ok"], env=_env(libdir), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "HOW yh_group_broadcast_weights (RCCL: ncclCommInitAll + grouped ncclBroadcast) over 3 devices" in r.stdout and "EQUAL True" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, "-c", GROUP_SCRIPT, pkg, "fail"], env=_env(libdir, RCCL_STANDIN_FAIL="initall"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "HOW through the host (RCCL path not taken: ncclCommInitAll: unhandled system error (stand-in librccl))" in r.stdout and "EQUAL True" in r.stdout, r.stdout


def test_bench_two_ranks_take_the_library_rccl_path(standin, tmp_path):
    """bench.py --gpus 2 with both ranks on cuda:0 (--rehearse-on-one-gpu: torch.distributed on gloo) and the library's own RCCL call
    against the stand-in (--rccl-library; a process with torch loaded already holds torch's bundled librccl under the soname, so the
    file is named by path): the record says the weights travelled by yh_rank_broadcast_weights, for the headline engine and for
    configs[4]'s."""
    _, so = standin
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--configs4-batch", "1", "--no-tflite", "--rehearse-on-one-gpu", "--rccl-library", so],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["weights_replication"].startswith("yh_rank_broadcast_weights (library RCCL"), b["config"]
    assert b["configs4"]["weights_replication"].startswith("yh_rank_broadcast_weights (library RCCL"), b["configs4"]["weights_replication"]
