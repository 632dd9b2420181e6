"""The scene back-end (shaders/pt_cloud.comp + pt_cloud_weights.comp, dispatched by
src/scene.rs:238-260). CPU part: the oracle (oracle/orc_scene.c) against hand-derived known answers of the frozen
deterministic reading. GPU part (-m gpu): the HIP kernels against the oracle, bit for bit, at the
reference's 640x480 frame ([80,60,1] x 8x8 dispatch) and at ragged small sizes. PARITY UNPINNED against the reference:
its shaders race and call pow() where GLSL leaves it undefined, and its run artefacts (map.bmp, depth.bmp) are lossy."""
import math

import numpy as np
import pytest


def _frame(rng, H, W, balls=True):
    depth = rng.integers(200, 4000, (H, W)).astype(np.uint16)
    depth[rng.random((H, W)) < 0.03] = 0                       # RealSense holes
    ci = np.zeros((H, W, 2), np.uint8)
    ci[H // 5:H // 3, W // 4:W // 2, 0] = 1                     # red robot
    ci[H // 2:H // 2 + H // 8, W // 2:W // 2 + W // 6, 0] = 2   # blue robot
    if balls:
        ci[H // 8:H // 8 + 6, W // 8:W // 8 + 7] = (3, 5)
        ci[H - 12:H - 5, W - 20:W - 9] = (3, 0)
        ci[3:5, 3:5] = (3, 200)                                 # id beyond the 100-entry ball table: ignored
    ci[0:2, :, 0] = 7                                           # an unknown class behaves as a robot (action_ty > 2 -> else branch)
    return depth, ci


def test_spec_log_against_libm(oracle):
    for x in (1.0, 1.4142135, 1.4142137, 9.0, 999.0, 4789.0, 0.3, 1e-3, 123.456, 3.0e38, 1.2e-38):
        assert abs(oracle.spec_logf(x) - math.log(np.float32(x))) <= 3e-7 * max(1.0, abs(math.log(np.float32(x)))), x


def test_scene_known_answers(oracle):
    """A flat terrain frame with one ball blob, derived by hand from the shader text."""
    H, W = 48, 64
    depth = np.full((H, W), 2000, np.uint16)
    ci = np.zeros((H, W, 2), np.uint8)
    ci[10:12, 20:24] = (3, 9)
    r = oracle.scene(depth, ci, mode=1)
    # ball 9: mean of new_pos = (x, H - int(H * depth * cos.. / 4000)) over its 8 pixels
    xs, ys = [], []
    for y in range(10, 12):
        for x in range(20, 24):
            ty = np.float32(0.55430907) * np.float32(y) * np.float32(2) / np.float32(H)
            tx = np.float32(0.9489646) * np.float32(x) * np.float32(2) / np.float32(W)
            d = np.float32(2000) * (np.float32(1) / np.sqrt(np.float32(1) + ty * ty)) * (np.float32(1) / np.sqrt(np.float32(1) + tx * tx))
            xs.append(x); ys.append(H - int(np.float32(H) * d / np.float32(4000)))
    assert r["balls"][9].tolist() == [np.float32(np.mean(xs)), np.float32(np.float64(sum(ys)) / 8), 8.0, 0.0]
    assert not r["balls"][np.arange(100) != 9].any()
    # world = (x, map, y, 0); the map's border is never written (loc.x > 0 && loc.x < width - 1, pt_cloud.comp:64)
    assert np.array_equal(r["world"][..., 0], np.tile(np.arange(W, dtype=np.float32), (H, 1)))
    assert np.array_equal(r["world"][..., 1], r["map"].astype(np.float32)) and np.array_equal(r["world"][..., 2], np.tile(np.arange(H, dtype=np.float32)[:, None], (1, W)))
    assert not r["map"][0].any() and not r["map"][-1].any() and not r["map"][:, 0].any() and not r["map"][:, -1].any()
    # a terrain bump stores at most val = the pixel's row index (pt_cloud.comp:116), so the map never exceeds H - 1; row 0 adds nothing
    assert 0 < r["map"].max() <= H - 1
    # SANE connections: conn1 = (down, down-left, left, up-left) neighbour distances, -1 off the frame
    y, x = 7, 9
    P = lambda yy, xx: np.array([xx, r["map"][yy, xx], yy], np.float32)
    for k, (dy, dx) in enumerate(((1, 0), (1, -1), (0, -1), (-1, -1))):
        dv = P(y, x) - P(y + dy, x + dx)
        assert r["conn1"][y, x, k] == np.sqrt(np.float32(dv[0] * dv[0] + dv[1] * dv[1]) + np.float32(dv[2] * dv[2]))
    assert r["conn1"][H - 1, 5, 0] == -1 and r["conn1"][4, 0, 1] == -1 and r["conn1"][4, 0, 2] == -1 and r["conn1"][0, 4, 3] == -1
    # conn0 = the same edges seen from the other end: (up, up-right, right, down-right)
    assert r["conn0"][y, x, 0] == r["conn1"][y - 1, x, 0] and r["conn0"][y, x, 1] == r["conn1"][y - 1, x + 1, 1]
    assert r["conn0"][y, x, 2] == r["conn1"][y, x + 1, 2] and r["conn0"][y, x, 3] == r["conn1"][y + 1, x + 1, 3]
    assert r["conn0"][0, 5, 0] == -1 and r["conn0"][5, W - 1, 2] == -1
    # STRICT: pack(x, y) = (x << 16) & y = 0, so every existing neighbour decodes to world(0, 0) (pt_cloud_weights.comp:32-42)
    s = oracle.scene(depth, ci, mode=0)
    assert np.array_equal(s["map"], r["map"]) and np.array_equal(s["balls"], r["balls"]) and np.array_equal(s["world"], r["world"])
    dv = P(y, x) - P(0, 0)
    want = np.sqrt(np.float32(dv[0] * dv[0] + dv[1] * dv[1]) + np.float32(dv[2] * dv[2]))
    assert (s["conn1"][y, x] == want).all() and s["conn1"][H - 1, 5, 0] == -1


def test_robot_bump_is_a_radial_sigmoid(oracle):
    """One robot pixel on an otherwise ball-class frame (balls add no bumps): the height map is the shader's sigmoid mound
    y = 100 / (1 + 999^(2 r / 20 - 1)) truncated to integers, r the distance to new_pos, inside its 40 x 40 window."""
    H, W = 96, 96
    depth = np.full((H, W), 1500, np.uint16)
    ci = np.zeros((H, W, 2), np.uint8)
    ci[..., 0] = 3
    ci[..., 1] = 150                                            # ball ids >= 100: ignored, no bump either
    yy, xx = 40, 50
    ci[yy, xx] = (1, 0)
    r = oracle.scene(depth, ci, mode=1)
    ty = 0.55430907 * yy * 2 / H; tx = 0.9489646 * xx * 2 / W
    d = 1500 / math.sqrt(1 + ty * ty) / math.sqrt(1 + tx * tx)
    py, px = H - int(H * d / 4000), xx
    m = r["map"]
    assert m[py, px] == int(100 / (1 + 999 ** -1.0))            # centre: r = 0
    for (dy, dx) in ((0, 5), (3, 4), (-7, 0), (10, -10), (-19, 19), (0, -20)):
        rr = math.hypot(dy, dx)
        want = 100 / (1 + 999 ** (2 * rr / 20 - 1))
        assert abs(int(m[py + dy, px + dx]) - want) <= 1.0, (dy, dx)   # (truncation of a float32 evaluation)
    assert m[py, px + 20] == 0 and m[py - 21, px] == 0          # outside the window [pos - 20, pos + 20)
    assert (m > 0).sum() <= 40 * 40


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,mode", [(480, 640, 0), (480, 640, 1), (37, 53, 1), (8, 8, 0), (100, 9, 1)])
def test_scene_hip_matches_oracle_bit_for_bit(built, oracle, H, W, mode):
    import yolact_amd as ya
    rng = np.random.default_rng(H * 1000 + W + mode)
    depth, ci = _frame(rng, H, W)
    sc = ya.Scene(W, H)
    sc.append(depth, ci, mode)
    got, want = sc.read(), oracle.scene(depth, ci, mode)
    for k in ("map", "balls", "world", "conn1", "conn0"):
        assert np.array_equal(got[k], want[k]), (k, H, W, mode)
    assert got["map"].max() > 0 and (H < 100 or got["balls"][5, 2] > 0)
    sc.append(depth, ci, mode)                                   # a second frame on the same handle: state is reset
    again = sc.read()
    assert all(np.array_equal(again[k], got[k]) for k in got)
    sc.close()


@pytest.mark.gpu
def test_scene_takes_the_class_image_from_a_classified_frame(built, oracle):
    """The consumer's view (src/scene.rs:91-93): STRICT reads the low 16 bits of the classified pixels - which classify leaves
    zero, so every pixel is terrain - SANE reads class and id where classify puts them (bits 31-24 / 23-16).
    The device-pointer form consumes yh_classify_frame_u32's device frame without a host round trip."""
    import yolact_amd as ya
    H, W = 480, 640
    rng = np.random.default_rng(3)
    depth, ci = _frame(rng, H, W)
    frame = (ci[..., 0].astype(np.uint32) << 24) | (ci[..., 1].astype(np.uint32) << 16)
    sc = ya.Scene(W, H)
    sc.append_classified(depth, frame_u32=frame, mode=ya.COMPAT_SANE)
    got, want = sc.read(), oracle.scene(depth, ci, 1)
    assert all(np.array_equal(got[k], want[k]) for k in want)
    sc.append_classified(depth, frame_u32=frame, mode=ya.COMPAT_STRICT)
    got, want = sc.read(), oracle.scene(depth, np.zeros_like(ci), 0)
    assert all(np.array_equal(got[k], want[k]) for k in want)
    # classify on the device, then the scene from the device-resident result
    y = ya.Yolact.init(seed=1, compat_mode=ya.COMPAT_SANE)
    cam = (rng.integers(0, 256, (H, W, 3), dtype=np.uint32) * np.array([1 << 24, 1 << 16, 1 << 8], np.uint32)).sum(-1).astype(np.uint32).reshape(-1)
    y.classify(cam)
    sc.append_classified(depth, frame_dev_ptr=y.interpreter.classify_device_frame(), mode=ya.COMPAT_SANE)
    got = sc.read()
    ci2 = np.stack([(cam.reshape(H, W) >> 24).astype(np.uint8), ((cam.reshape(H, W) >> 16) & 255).astype(np.uint8)], -1)
    want = oracle.scene(depth, ci2, 1)
    assert all(np.array_equal(got[k], want[k]) for k in want)
    print(f"scene 640x480: {sc.time(20):.3f} ms per frame")
    sc.close()
