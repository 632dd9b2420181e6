"""The oracle's network primitives pinned against torch CPU (fp32), as SURVEY.md §8c prescribes:
the network itself is SPEC-EXTERNAL (parity unpinned against the reference), its building blocks
are not. CPU only, small shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res,act", [
    (1, 9, 11, 8, 16, 3, 1, 1, False, 0),
    (2, 12, 10, 16, 24, 3, 2, 1, True, 1),
    (1, 7, 7, 32, 20, 1, 1, 0, True, 1),
    (1, 21, 19, 3, 8, 7, 2, 3, False, 1),
    (1, 6, 6, 16, 12, 1, 2, 0, False, 0),
])
def test_conv2d(oracle, n, h, w, cin, cout, k, stride, pad, res, act):
    g = torch.Generator().manual_seed(n * 1000 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (k * k * cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y = F.conv2d(x, wt, b, stride=stride, padding=pad)
    r = torch.randn(y.shape, generator=g) if res else None
    if res:
        y = y + r
    if act == 1:
        y = F.relu(y)
    got = oracle.conv2d(_nhwc(x), wt.permute(0, 2, 3, 1).contiguous().numpy(), b.numpy(), stride, pad,
                        _nhwc(r) if res else None, act, f16=False, nthreads=2)
    assert np.allclose(got, _nhwc(y), rtol=1e-4, atol=1e-5)


def test_conv2d_tanh(oracle):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 5, 5, generator=g)
    wt = torch.randn(8, 16, 3, 3, generator=g) / 12
    b = torch.zeros(8)
    y = torch.tanh(F.conv2d(x, wt, b, padding=1))
    got = oracle.conv2d(_nhwc(x), wt.permute(0, 2, 3, 1).contiguous().numpy(), b.numpy(), 1, 1, None, 2, f16=False, nthreads=1)
    assert np.allclose(got, _nhwc(y), atol=2e-6)


@pytest.mark.parametrize("h,w,ho,wo", [(18, 18, 35, 35), (35, 35, 69, 69), (5, 7, 10, 14), (9, 9, 9, 9), (4, 4, 7, 5)])
def test_bilinear_matches_torch_align_corners_false(oracle, h, w, ho, wo):
    x = torch.randn(2, 8, h, w, generator=torch.Generator().manual_seed(h * w))
    y = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
    got = oracle.bilinear(_nhwc(x), ho, wo)
    assert np.allclose(got, _nhwc(y), atol=1e-5)


def test_maxpool(oracle):
    for h, w in ((275, 13), (12, 12), (7, 9)):
        x = torch.randn(1, 8, h, w, generator=torch.Generator().manual_seed(h))
        y = F.max_pool2d(x, 3, 2, 1)
        assert np.array_equal(oracle.maxpool3x3s2(_nhwc(x)), _nhwc(y))


def test_f16_rounding_is_ieee_rne(oracle):
    r = np.random.default_rng(0)
    v = np.concatenate([r.normal(0, 1, 4000), r.normal(0, 1e-6, 1000), r.normal(0, 2e4, 1000),
                        [65504, 65519.9, 65520, 6e-8, 2.9802322e-8, 3e-8, 0.0, -0.0]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = v.astype(np.float16).astype(np.float32)
    assert np.array_equal(oracle.f16_round(v), want)


def test_spec_exp_and_tanh_accuracy(oracle):
    xs = np.linspace(-30, 10, 2001)
    rel = max(abs(oracle.spec_expf(x) - np.exp(x)) / np.exp(x) for x in xs)
    assert rel < 2e-6
    assert max(abs(oracle.spec_tanhf(x) - np.tanh(x)) for x in np.linspace(-9, 9, 1801)) < 3e-7


def test_softmax_and_detect_against_torch_reference(oracle):
    """Tail semantics (softmax -> threshold -> per-class top-k -> Fast-NMS -> top-N) re-derived with
    torch ops on a small problem; class ids / priors must agree, scores and boxes to fp tolerance."""
    rng = np.random.default_rng(3)
    P, C, hp = 300, 9, 16
    loc = rng.normal(0, 0.5, (P, 4)).astype(np.float32)
    conf = rng.normal(0, 2.0, (P, C)).astype(np.float32)
    conf[:, 0] += 2.0
    mask = np.tanh(rng.normal(0, 1, (P, 32))).astype(np.float32)
    proto = np.maximum(rng.normal(0, 1, (hp, hp, 32)), 0).astype(np.float32)
    pri = np.stack([rng.uniform(0.1, 0.9, P), rng.uniform(0.1, 0.9, P), rng.uniform(0.05, 0.3, P), rng.uniform(0.05, 0.3, P)], 1).astype(np.float32)
    dets, masks = oracle.detect(loc, conf, mask, proto, pri, num_classes=C, top_k=20, max_dets=15, conf_thresh=0.05, nms_thresh=0.5)
    # torch restatement
    tl, tp = torch.tensor(loc), torch.tensor(pri)
    prob = torch.softmax(torch.tensor(conf), 1)
    cxcy = tp[:, :2] + tl[:, :2] * 0.1 * tp[:, 2:]
    wh = tp[:, 2:] * torch.exp(tl[:, 2:] * 0.2)
    boxes = torch.cat([cxcy - wh / 2, cxcy + wh / 2], 1)
    surv = []
    for c in range(1, C):
        s = prob[:, c]
        idx = torch.nonzero(s > 0.05).flatten()
        order = sorted(idx.tolist(), key=lambda p: (-s[p].item(), p))[:20]
        bb = boxes[order]
        for j, pj in enumerate(order):
            keep = True
            for i in range(j):
                a, b = bb[i], bb[j]
                iw = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0)
                ih = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
                inter = iw * ih
                uni = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
                if uni > 0 and inter / uni > 0.5:
                    keep = False
            if keep:
                surv.append((-s[pj].item(), c - 1, j, pj))
    surv.sort()
    surv = surv[:15]
    assert [(d["class_id"], d["prior"]) for d in dets] == [(c, p) for _, c, _, p in surv]
    for d, (ns, c, _, p) in zip(dets, surv):
        assert abs(d["score"] + ns) < 1e-6
        assert np.allclose(d["box"], boxes[p].numpy(), atol=1e-5)
    # masks: sigmoid(proto @ coeff) > 0.5 inside the padded crop window
    for k, d in enumerate(dets):
        lin = torch.tensor(proto) @ torch.tensor(mask[d["prior"]])
        m = (torch.sigmoid(lin) > 0.5).numpy()
        x1, y1, x2, y2 = [v * hp for v in d["box"]]
        xa, xb = max(min(x1, x2) - 1, 0), min(max(x1, x2) + 1, hp)
        ya, yb = max(min(y1, y2) - 1, 0), min(max(y1, y2) + 1, hp)
        xs, ys = np.arange(hp)[None, :], np.arange(hp)[:, None]
        want = m & (xs >= xa) & (xs < xb) & (ys >= ya) & (ys < yb)
        agree = (want == masks[k].astype(bool)) | (np.abs(lin.numpy()) < 1e-5)
        assert agree.all()


def test_network_shapes_and_flops(oracle):
    net = oracle.Net(50, 550, 81, seed=1)
    assert net.P == 19248 and (net.hp, net.wp) == (138, 138)       # SURVEY.md Appendix B sanity check
    assert abs(net.flops_per_frame() / 1e9 - 118.28) < 0.05        # 118.28 GFLOP
    net101 = oracle.Net(101, 700, 81, seed=1)
    assert net101.P == 30963 and net101.hp == 176
    assert abs(net101.flops_per_frame() / 1e9 - 262.93) < 0.1
