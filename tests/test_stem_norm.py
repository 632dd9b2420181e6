"""The fused stem normalises raw bytes as (v - mean) * (1 / std) in f32, rounded to f16 (conv_igemm.hip: stem_pool_f16,
store_patch); the spec and the oracle (oracle/orc_net.c preprocessing) say (v - mean) / std. Both give the same
f16 for every byte value of every channel - checked here exhaustively, so the kernel's form is not a tolerance."""
import numpy as np

MEAN = np.array([123.68, 116.78, 103.94], np.float32)
STD = np.array([58.40, 57.12, 57.38], np.float32)


def test_reciprocal_form_equals_division_for_all_768_inputs():
    v = np.arange(256, dtype=np.float32)
    for c in range(3):
        ref = ((v - MEAN[c]) / STD[c]).astype(np.float32).astype(np.float16)
        r = (np.float32(1.0) / STD[c]).astype(np.float32)
        alt = ((v - MEAN[c]) * r).astype(np.float32).astype(np.float16)
        assert np.array_equal(ref.view(np.uint16), alt.view(np.uint16)), c
