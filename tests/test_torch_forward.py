"""The whole network of the oracle (oracle/orc_net.c, f32 mode) against a plain torch-CPU forward of the same layer
table on the same seeded weights (bench.torch_cpu_forward, which bench.py also times as the stronger CPU baseline):
two independent implementations of DESIGN.md §2 - conv order and strides, residual wiring, FPN top-down adds,
bilinear align_corners=False, the shared head and its (level, y, x, anchor) prior order - must agree to f32 rounding.
This pins the oracle's NETWORK against torch's own convolutions; it is still not parity with the reference's tflite
interpreter (model file absent: parity unpinned, oracle/oracle.h)."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("backbone,S", [(50, 96), (101, 65)])
def test_oracle_network_matches_torch_cpu(oracle, backbone, S):
    import bench
    net = oracle.Net(backbone, S, 81, seed=3)
    img = np.random.default_rng(backbone).integers(0, 256, (2, S, S, 3), dtype=np.uint8)
    want = net.forward(img, f16=False)
    with torch.no_grad():
        got = bench.torch_cpu_forward(torch, bench.parse_blob(net.blob), img, backbone)
    for name, a, b in zip(("loc", "conf", "mask", "proto"), got, want):
        a = a.numpy()
        assert a.shape == b.shape, name
        assert np.abs(a - b).max() <= 2e-4 * max(1.0, np.abs(b).max()), (name, float(np.abs(a - b).max()))
