"""-m gpu: the compiled (C++) host-side mirror of src/yolact.rs (tiny-object-detection_amd/host/)
run as a separate executable, the way src/scene.rs drives Yolact: init once, classify a packed
frame in place. Must agree bit for bit with the Python mirror (same C ABI underneath)."""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tiny-object-detection_amd", "lib", "yolact_demo")


def _frame(oracle, golden_dir, w, h):
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(golden_dir, "red_robot.png")).convert("RGB").resize((w, h), Image.BILINEAR))
    return oracle.pack_rgb(rgb)


def test_cpp_host_matches_python_mirror(built, oracle, golden_dir, tmp_path):
    import yolact_amd as ya
    assert os.path.exists(DEMO), "make -C tiny-object-detection_amd builds lib/yolact_demo"
    frame = _frame(oracle, golden_dir, 640, 480)
    fin, fout = tmp_path / "in.u32", tmp_path / "out.u32"
    frame.tofile(fin)
    r = subprocess.run([DEMO, str(fin), str(fout), "640", "480", "224", str(ya.COMPAT_SANE)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "gfx950" in r.stdout and "low-16-bit non-zero: 0" in r.stdout        # scene.rs:62 banner; A10
    got = np.fromfile(fout, np.uint32)
    y = ya.Yolact.init(seed=1, input_size=224, compat_mode=ya.COMPAT_SANE)
    want = frame.copy()
    y.classify(want)
    y.interpreter.close()
    assert np.array_equal(got, want)


def test_cpp_host_with_a_tflite_model_and_errors(built, oracle, golden_dir, tmp_path):
    import tfl_builder as B
    import tfl_models as M
    import yolact_amd as ya
    model = tmp_path / "FRC_model.tflite"
    model.write_bytes(B.serialize(M.mobilenet_like(np.random.default_rng(5), S=64, C=6)))
    frame = _frame(oracle, golden_dir, 160, 120)
    fin, fout = tmp_path / "in.u32", tmp_path / "out.u32"
    frame.tofile(fin)
    r = subprocess.run([DEMO, str(fin), str(fout), "160", "120", "64", str(ya.COMPAT_SANE), str(model)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    eng = ya.TfliteEngine(model.read_bytes())
    want = frame.copy()
    eng.classify_frame(want, 160, 120, ya.COMPAT_SANE)
    eng.close()
    assert np.array_equal(np.fromfile(fout, np.uint32), want)
    # the reference .expect()s: a missing model file "panics" with the reference's message
    r = subprocess.run([DEMO, str(fin), str(fout), "160", "120", "64", "1", str(tmp_path / "missing.tflite")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "failed to load model" in r.stderr
