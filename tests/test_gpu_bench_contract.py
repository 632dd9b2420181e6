"""-m gpu: bench.py's output contract (the driver parses exactly one JSON line from it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--cpu-budget", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "frames/s" and b["n_gpus"] == 1 and b["steps"] == 3 and b["warmup"] == 1
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["dtype"] == "f16" and b["data"] == "synthetic" and "workload" in b["config"] and "model" not in b["config"]
    assert b["value"] > 0 and abs(b["value"] - 8 * 3 / (b["ms_per_step"] * 3e-3)) < 0.01 * b["value"]
    rf = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    acc = cb["engine_vs_oracle_frc_balls"]                        # the restated acceptance target, over ALL detections of both sides
    assert acc["mask_iou_all"] >= 0.99 and acc["unmatched_oracle"] == 0 and acc["unmatched_engine"] == 0 and acc["oracle_dets"] >= 5, acc
    acc = cb["engine_vs_oracle_same_frame"]                       # a noise frame fills the list: ties at the top-k cut may swap
    assert acc["unmatched_oracle"] <= 3 and acc["unmatched_engine"] <= 3 and acc["unmatched_above_cut"] == 0 and acc["mask_iou_above_cut"] >= 0.99, acc
    assert b["batch1"]["value"] > 0
    # BASELINE.json configs[4] in the same line: YOLACT-700 R101, fp8, this GPU's share (8 frames) of 64 frames over 8 GPUs
    c4 = b["configs4"]
    assert c4["value"] > 0 and c4["dtype"] == "fp8" and "R101" in c4["workload"] and abs(c4["value"] - 8 * c4["steps"] / (c4["ms_per_step"] * c4["steps"] * 1e-3)) < 0.01 * c4["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c4["roofline"], k
    assert c4["fp8_launches"]["launches"] == 36 and c4["fp8_launches"]["peak"] == 5000.0 and 0 < c4["fp8_launches"]["frac"] < 1
    for k in ("engine_vs_oracle_fp8_mode", "engine_fp8_vs_oracle_f16"):
        assert c4["accuracy"][k]["engine_dets"] > 0 and "matched_class_and_prior" in c4["accuracy"][k], k


def test_graft_entry_smoke_runs(built):
    """__graft_entry__.smoke() is what the driver runs on the GPU box before the bench."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


def test_two_rank_control_flow_on_one_gpu(built):
    """The N > 1 path of bench.py (barriers, MAX over ranks, rank-0-only extras, weight broadcast) with two
    ranks sharing cuda:0 over gloo (--rehearse-on-one-gpu): a control-flow rehearsal, not a measurement.
    The real N = 2/4/8 runs use RCCL with one rank per GPU and are the driver's."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--rehearse-on-one-gpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["global_batch"] == 8 and "cpu_baseline" not in b
    assert abs(b["value"] - 2 * 4 * 3 / (b["ms_per_step"] * 3e-3)) < 0.01 * b["value"]
