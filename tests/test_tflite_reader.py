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
