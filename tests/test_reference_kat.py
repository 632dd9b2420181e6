"""Oracle (oracle/orc_ref.c) pinned against the known-answer vectors of the reference's own logic
(tests/golden/reference_kat.json). CPU only."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "reference_kat.json")) as f:
        return json.load(f)


def test_gated_argmax(oracle, kat):
    rows = kat["gated_argmax"]
    dets = np.zeros((len(rows), 81), np.float32)
    for i, r in enumerate(rows):
        dets[i, :4] = [float(v) for v in r["in"]]
    dets[:, 4:] = 99.0  # logits 4..80 are ignored (yolact.rs:110 `.take(4)`)
    got = oracle.gated_argmax(dets)
    assert got.tolist() == [r["cls"] for r in rows]


def test_appendix_a1_table(oracle):
    nan = float("nan")
    table = [([0.5, 9, 9, 9], 0), ([-1, 2, 2, 1], 1), ([-1, 1, 2, 3], 3), ([-1, 3, 2, 1], 1), ([-1, 1, 3, 2], 2),
             ([-1, -1, -1, -1], 0), ([0, 0, 0, 0], 0), ([-1, nan, 1, 0.5], 2)]
    dets = np.zeros((len(table), 81), np.float32)
    for i, (v, _) in enumerate(table):
        dets[i, :4] = v
    assert oracle.gated_argmax(dets).tolist() == [c for _, c in table]


def test_pixel_packing(oracle, kat):
    rgb = np.array([r["rgb"] for r in kat["pixel"]], np.uint8)
    want = np.array([r["u32"] for r in kat["pixel"]], np.uint32)
    assert np.array_equal(oracle.pack_rgb(rgb), want)
    assert np.array_equal(oracle.unpack_rgb(want).reshape(-1, 3), rgb)
    assert oracle.pack_rgb(np.array([18, 52, 86], np.uint8))[0] == 0x12345600  # Appendix A.4


def test_dequant(oracle, kat):
    for r in kat["dequant"]:
        got = oracle.dequant_u8(np.array([r["x"]], np.uint8), r["scale"], r["zp"])[0]
        assert got == np.float32(r["out"]), r
    assert oracle.dequant_u8(np.array([130], np.uint8), 0.5, 128)[0] == 1.0       # Appendix A.5
    assert oracle.dequant_u8(np.array([0], np.uint8), 0.0625, 128)[0] == -8.0


def test_flood_fill(oracle, kat):
    for g in kat["flood_fill"]:
        rc, ids = oracle.terrible_id(np.array(g["classes"], np.uint8), 28)
        assert bool(rc) == g["diverges"], g["name"]
        if not g["diverges"]:
            assert ids.tolist() == g["ids"], g["name"]
            assert set(g["ids"]) == {-1}  # Appendix A.6: a terminating run labels nothing


def test_pack_and_upsample(oracle, kat):
    # A.2 through postprocess: cls from logits, ids all -1 in any terminating run -> cls << 24
    for cls in range(4):
        want = [p["u32"] for p in kat["pack"] if p["cls"] == cls and p["id"] == -1][0]
        cells = np.full((784, 81), -1.0, np.float32)
        cell = 5 * 28 + 9
        if cls:
            cells[cell, cls] = 2.0
        rc, out = oracle.postprocess_tile(cells, 28, 81, 0)
        assert rc == 0
        img = out.reshape(224, 224)
        assert (img[40:48, 72:80] == want).all()          # A.3: rows 8r..8r+7, cols 8c..8c+7
        assert (np.delete(img, np.s_[40:48], 0) == 0).all()
    # labelled ball cells would pack to 0 (A8'): checked at the expression level by the KAT table
    for p in kat["pack"]:
        assert p["u32"] == ((p["cls"] << 24) & ((p["id"] & 0xFFFFFFFF) << 16) & 0xFFFFFFFF)


def test_strict_divergence_and_sane_mode(oracle):
    cells = np.full((784, 81), -1.0, np.float32)
    cells[100, 3] = 1.0
    cells[101, 3] = 1.0
    rc, _ = oracle.postprocess_tile(cells, 28, 81, 0)
    assert rc == 1  # two 4-adjacent ball cells: the reference loops forever
    rc, out = oracle.postprocess_tile(cells, 28, 81, 1)
    assert rc == 0
    img = out.reshape(224, 224)
    assert img[3 * 8, (100 - 84) * 8] == (3 << 24) | (0 << 16)
    assert img[0, 0] == 0xFF << 16  # background: cls 0, id -1 as a byte


def test_consumer_low16(oracle, kat):
    v = np.array([r["u32"] for r in kat["consumer_low16"]], np.uint32)
    assert oracle.consumer_low16(v).tolist() == [r["u16"] for r in kat["consumer_low16"]]


def test_triangle_resize_properties(oracle, golden_dir):
    """image 0.24.1 is not vendored (parity unpinned): check the algorithm's invariants instead."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(golden_dir, "frc_balls.png")).convert("RGB"))
    assert np.array_equal(oracle.resize_triangle_rgb8(img, 224, 224), img)  # same size: copy
    flat = np.full((48, 64, 3), 77, np.uint8)
    assert (oracle.resize_triangle_rgb8(flat, 45, 22) == 77).all()          # weights sum to 1
    assert (oracle.resize_triangle_rgb8(flat, 130, 97) == 77).all()
    # exact 2x upscale of a 1-D ramp stays within the source range and is monotone
    ramp = np.repeat(np.arange(0, 200, 10, dtype=np.uint8)[None, :, None], 3, 2).repeat(4, 0)
    up = oracle.resize_triangle_rgb8(ramp, 40, 4)[0, :, 0].astype(int)
    assert (np.diff(up) >= 0).all() and up.min() >= 0 and up.max() <= 190
    # downscale by an integer factor of a constant-per-block image reproduces block means (+-1)
    blocks = np.kron(np.arange(16, dtype=np.uint8).reshape(4, 4) * 10, np.ones((8, 8), np.uint8))
    down = oracle.resize_triangle_rgb8(np.repeat(blocks[:, :, None], 3, 2), 4, 4)[:, :, 0].astype(int)
    assert np.abs(down - np.arange(16).reshape(4, 4) * 10).max() <= 12  # triangle support spans neighbours


def test_classify_pre_post_shapes(oracle, golden_dir):
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(golden_dir, "red_robot.png")).convert("RGB"))
    frame_rgb = np.asarray(Image.fromarray(img).resize((640, 480), Image.BILINEAR))
    frame = oracle.pack_rgb(frame_rgb)
    tiles = oracle.classify_pre(frame, 640, 480, 224)
    sq = oracle.resize_triangle_rgb8(frame_rgb, 448, 224)
    assert np.array_equal(tiles[0], sq[:, :224]) and np.array_equal(tiles[1], sq[:, 224:])  # yolact.rs:213-214
    cells = np.full((2, 784, 81), -1.0, np.float32)
    cells[0, :28, 1] = 1.0   # top cell row of tile 0: red robot
    rc, out = oracle.classify_post(cells, 224, 81, 0, 640, 480)
    assert rc == 0
    out = out.reshape(480, 640)
    assert (out & 0x00FFFFFF == 0).all() and out[0, 0] >> 24 == 1 and out[479, 639] == 0
    assert (oracle.consumer_low16(out) == 0).all()  # A10: the consumer always reads 0
