"""CPU: the oracle's OCP FP8 E4M3 encoder/decoder (oracle/orc_fp8.c; groundwork for the fp8 convolution
path of DESIGN.md §10) pinned against the format definition itself: the decode table is built here from
the bit fields, and the encoder must pick, for EVERY f16 input, the nearest table value with ties to the
even code, saturating at +-448."""
import numpy as np


def _table():
    t = np.zeros(256, np.float64)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        v = np.nan if (e == 15 and m == 7) else (m / 8 * 2.0 ** -6 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
        t[b] = -v if s else v
    return t


def test_decode_table_matches_the_format(oracle):
    got = oracle.e4m3_decode_table().astype(np.float64)
    want = _table()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    assert want[0x7E] == 448.0 and want[0x01] == 2.0 ** -9 and want[0x08] == 2.0 ** -6


def test_encode_is_nearest_even_saturating_for_every_f16(oracle):
    x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    fin = np.isfinite(x)
    codes = oracle.quantize_e4m3(x)
    t = _table()
    pos = t[:0x7F]                                            # 0 .. 448, increasing, no NaN
    a = np.abs(x[fin]).astype(np.float64)
    d = np.abs(a[:, None] - pos[None, :])
    best = d.min(axis=1)
    cand = d == best[:, None]                                 # one or two nearest codes
    lo = cand.argmax(axis=1)
    hi = 0x7E - cand[:, ::-1].argmax(axis=1)
    want = np.where(lo == hi, lo, np.where(lo % 2 == 0, lo, hi))   # tie -> even code (adjacent codes differ in the mantissa LSB)
    want = np.where(a >= 448.0, 0x7E, want)
    sign = (np.signbit(x[fin]).astype(np.uint8) << 7)
    assert np.array_equal(codes[fin], (want.astype(np.uint8) | sign))
    # infinities saturate, NaN stays NaN
    assert codes[x == np.inf][0] == 0x7E and codes[x == -np.inf][0] == 0xFE
    assert np.all((codes[np.isnan(x)] & 0x7F) == 0x7F)


def test_round_trip_and_scale(oracle):
    t = oracle.e4m3_decode_table()
    fin = ~np.isnan(t)
    codes = oracle.quantize_e4m3(t[fin])
    assert np.array_equal(oracle.e4m3_decode_table()[codes], t[fin])          # decode(encode(v)) == v for every finite code
    x = np.array([1.0, 3.0, 100.0, -0.3], np.float32)
    assert np.array_equal(oracle.quantize_e4m3(x, 0.5), oracle.quantize_e4m3(x * np.float32(0.5)))


def test_fp8_study_modes_of_the_net_forward(oracle):
    """The oracle's fp8 accuracy study: E4M3-rounded operands on the K-heavy 3x3 convs move the
    heads by a few per cent and nothing else; the protonet-only mode leaves the prediction heads untouched."""
    S = 128
    net = oracle.Net(50, S, 81, seed=1)
    x = np.random.default_rng(0).integers(0, 256, (1, S, S, 3), dtype=np.uint8)
    ref = net.forward(x, f16=True)
    rel = lambda u, v: float(np.sqrt(((u - v) ** 2).mean()) / np.sqrt((u ** 2).mean()))
    for mode in (1, 2):
        got = net.forward(x, f16=True, fp8_study=mode)
        errs = [rel(u, v) for u, v in zip(ref, got)]
        assert all(0.005 < e < 0.25 for e in errs), (mode, errs)
    got = net.forward(x, f16=True, fp8_study=3)
    assert all(np.array_equal(u, v) for u, v in zip(ref[:3], got[:3])) and 0.005 < rel(ref[3], got[3]) < 0.25
    again = net.forward(x, f16=True)
    assert all(np.array_equal(u, v) for u, v in zip(ref, again))          # the switch does not stick
