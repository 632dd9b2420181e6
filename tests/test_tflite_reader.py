"""CPU: the product's TFLite flatbuffer reader (csrc/tflite_model.h, via yh_tfl_validate) against
models written by tests/tfl_builder.py, plus robustness: truncations and byte flips of a valid
file must be rejected or accepted — never crash or read out of bounds."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfl_builder as B
import tfl_models as M


def test_reader_accepts_builder_output(built):
    import yolact_amd as ya
    rng = np.random.default_rng(0)
    m = M.mobilenet_like(rng)
    blob = B.serialize(m)
    ok, nt, no, msg = ya.tfl_validate(blob)
    assert ok, msg
    assert nt == len(m.tensors) and no == len(m.ops)
    for code in ("CONV_2D", "DEPTHWISE_CONV_2D", "ADD", "PAD", "RESIZE_BILINEAR", "TANH", "RELU", "QUANTIZE",
                 "CONCATENATION", "RESHAPE", "DEQUANTIZE"):
        ok, nt, no, msg = ya.tfl_validate(B.serialize(M.single_op(code, rng)))
        assert ok and no == 1, (code, msg)


def test_reader_rejects_garbage_without_crashing(built):
    import yolact_amd as ya
    rng = np.random.default_rng(1)
    blob = B.serialize(M.mobilenet_like(rng))
    assert not ya.tfl_validate(b"")[0] and not ya.tfl_validate(b"\x00" * 64)[0]
    assert not ya.tfl_validate(blob[:8])[0]
    for cut in list(range(8, 400, 7)) + list(range(400, len(blob), 997)):
        ya.tfl_validate(blob[:cut])                      # must return, ok or not
    for trial in range(300):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, min(len(b), 6000)))] = int(rng.integers(0, 256))   # hit the metadata, not weights
        ya.tfl_validate(bytes(b))
    bad = bytearray(blob)
    bad[4:8] = b"XXXX"
    ok, _, _, msg = ya.tfl_validate(bytes(bad))
    assert not ok and "TFL3" in msg


def _structural_mutants(rng):
    """(description, model) pairs: builder models broken in exactly the ways the plan builder (tflite_exec.hip:
    prepare) would otherwise trip over: operand indices, arity, ranks, zero strides / dilations / multipliers."""
    out = []

    def mut(code, what, fn, **kw):
        m = M.single_op(code, rng, **kw)
        fn(m)
        out.append((f"{code}: {what}", m))
    for code in ("CONV_2D", "DEPTHWISE_CONV_2D"):
        mut(code, "input index -1", lambda m: m.ops[0].inputs.__setitem__(0, -1))
        mut(code, "weight index -1", lambda m: m.ops[0].inputs.__setitem__(1, -1))
        mut(code, "bias index -7", lambda m: m.ops[0].inputs.__setitem__(2, -7))
        mut(code, "one input", lambda m: m.ops[0].__setattr__("inputs", m.ops[0].inputs[:1]))
        mut(code, "no output", lambda m: m.ops[0].__setattr__("outputs", []))
        mut(code, "stride_h 0", lambda m: m.ops[0].opts.__setitem__("stride_h", 0))
        mut(code, "stride_w -2", lambda m: m.ops[0].opts.__setitem__("stride_w", -2))
        mut(code, "dilation 0", lambda m: m.ops[0].opts.__setitem__("dil_h", 0))
        mut(code, "3-D input", lambda m: m.tensors[0].__setattr__("shape", m.tensors[0].shape[1:]))
        mut(code, "3-D output", lambda m: m.tensors[-1].__setattr__("shape", m.tensors[-1].shape[1:]))
    mut("DEPTHWISE_CONV_2D", "depth multiplier 0", lambda m: m.ops[0].opts.__setitem__("depth_multiplier", 0))
    for code in ("ADD", "TANH", "RELU", "QUANTIZE", "DEQUANTIZE", "PAD", "RESIZE_BILINEAR", "CONCATENATION", "RESHAPE"):
        mut(code, "no output", lambda m: m.ops[0].__setattr__("outputs", []))
        mut(code, "input index -1", lambda m: m.ops[0].inputs.__setitem__(0, -1))
        mut(code, "no input", lambda m: m.ops[0].__setattr__("inputs", []))
    mut("PAD", "3-D input", lambda m: m.tensors[0].__setattr__("shape", m.tensors[0].shape[1:]))
    mut("PAD", "3-D output", lambda m: m.tensors[2].__setattr__("shape", m.tensors[2].shape[1:]))
    mut("RESIZE_BILINEAR", "2-D output", lambda m: m.tensors[2].__setattr__("shape", m.tensors[2].shape[2:]))
    mut("CONCATENATION", "rank mismatch", lambda m: m.tensors[0].__setattr__("shape", m.tensors[0].shape[1:]))
    return out


def test_graph_validation_rejects_what_the_plan_builder_would_trip_over(built, tmp_path):
    """yh_tfl_validate (no GPU) now also checks everything prepare() dereferences. Every structural
    mutant is rejected with a message; the same files, plus byte-flip and truncation mutants, go through the reader
    compiled with AddressSanitizer + UBSan (tests/tfl_reader_harness.cpp), which must finish without a report."""
    import subprocess
    import yolact_amd as ya
    rng = np.random.default_rng(11)
    files = []
    for i, (what, m) in enumerate(_structural_mutants(rng)):
        blob = bytes(B.serialize(m))
        ok, _, _, msg = ya.tfl_validate(blob)
        assert not ok and msg, what
        p = tmp_path / f"s{i}.tflite"
        p.write_bytes(blob)
        files.append(str(p))
    # forged vector lengths and random metadata damage
    good = bytes(B.serialize(M.mobilenet_like(rng)))
    for i in range(150):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, min(len(b), 6000)))
            b[at:at + 4] = int(rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0, int(rng.integers(0, 1 << 32))])).to_bytes(4, "little")
        ya.tfl_validate(bytes(b))                        # must return
        p = tmp_path / f"f{i}.tflite"
        p.write_bytes(bytes(b[:len(good)]))
        files.append(str(p))
    (tmp_path / "good.tflite").write_bytes(good)
    exe = tmp_path / "harness"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(root, "tiny-object-detection_amd", "csrc"), "-o", str(exe),
                           os.path.join(root, "tests", "tfl_reader_harness.cpp")])
    r = subprocess.run([str(exe), str(tmp_path / "good.tflite")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert lines[0].endswith("good.tflite ok [%s]" % lines[0].split("[")[-1].rstrip("]")) and " ok" in lines[0]
    assert sum("rejected" in ln for ln in lines) >= len(_structural_mutants(rng))


def test_oracle_fixed_point_primitives():
    """gemmlowp primitives against exact rational arithmetic."""
    import tfl_oracle as O
    from fractions import Fraction
    rng = np.random.default_rng(2)
    for _ in range(200):
        real = float(rng.uniform(1e-4, 0.9999))
        m, s = O.quantize_multiplier(real)
        assert (1 << 30) <= m < (1 << 31) and abs(m * 2.0 ** (s - 31) - real) < real * 2e-9
        x = rng.integers(-(1 << 24), 1 << 24, 50)
        got = O.mbqm(x, m, s)
        want = np.array([float(Fraction(int(v)) * Fraction(m, 1 << 31) * Fraction(2) ** s) for v in x])
        assert np.abs(got - want).max() <= 1.0 + 1e-6        # two roundings: within one unit of exact
    assert O.mbqm(np.array([100]), 1 << 30, 1)[0] == 100     # 0.5 * 2^1
    assert O._rdbpot(np.array([5, -5, 6, -6, 7]), 2).tolist() == [1, -1, 2, -2, 2]   # ties away from zero


def test_oracle_scales_are_float32_like_the_file():
    """A .tflite stores scales as float32 and TFLite promotes that value to double; the oracle must
    not see more precision than the file carries (it once did: 1-LSB differences against the GPU
    executor on 2 of 150 528 elements of a 1x1 conv)."""
    import tfl_oracle as O
    sx, sw, so = 0.043721839, 0.0013810679, 0.027364615      # doubles that are not float32 values
    for s in (sx, sw, so):
        assert float(np.float32(s)) != s
    a = O.quantize_multiplier(O._f32(sx) * O._f32(sw) / O._f32(so))
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (1, 6, 6, 8), dtype=np.uint8)
    w = rng.integers(0, 256, (4, 1, 1, 8), dtype=np.uint8)
    b = rng.integers(-100, 100, 4).astype(np.int32)
    y1 = O.conv2d_u8(x, 120, sx, w, 128, sw, b, 110, so, (1, 1), 0, 0)
    y2 = O.conv2d_u8(x, 120, float(np.float32(sx)), w, 128, float(np.float32(sw)), b, 110, float(np.float32(so)), (1, 1), 0, 0)
    assert np.array_equal(y1, y2) and a == O.quantize_multiplier(float(np.float32(sx)) * float(np.float32(sw)) / float(np.float32(so)))
