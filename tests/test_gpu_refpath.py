"""-m gpu: the reference's own pre/post-processing (src/yolact.rs:52-131, :192-234) on device,
bit for bit against oracle/orc_ref.c and the known-answer vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=224, max_batch=2, use_graph=True)
    e.load_weights(e.generate_weights(seed=1))
    yield e
    e.close()


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 448, 224), (448, 224, 640, 480), (224, 224, 224, 224),
                                         (320, 240, 448, 224), (33, 17, 91, 5), (100, 100, 7, 3)])
def test_triangle_resize_bit_exact(eng, oracle, sw, sh, dw, dh):
    src = np.random.default_rng(sw + dh).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    assert np.array_equal(eng.resize_triangle(src, dw, dh), oracle.resize_triangle_rgb8(src, dw, dh))


def test_postprocess_known_answers(eng, golden_dir):
    with open(os.path.join(golden_dir, "reference_kat.json")) as f:
        kat = json.load(f)
    rows = kat["gated_argmax"][:784]
    cells = np.full((1, 784, 81), 50.0, np.float32)   # logits 4..80 ignored
    for i, r in enumerate(rows):
        cells[0, i, :4] = [float(v) for v in r["in"]]
    import yolact_amd as ya
    try:
        out = eng.postprocess_cells(cells, ya.COMPAT_STRICT)[0]
        got = (out[::8, ::8].reshape(-1) >> 24).tolist()
        assert got == [r["cls"] for r in rows]
        assert (out & 0x00FFFFFF == 0).all()
    except ya.YhError as e:                          # adjacent class-3 cells in the KAT order: diverges
        assert e.code == ya.EDIVERGE
        out = eng.postprocess_cells(cells, ya.COMPAT_SANE)[0]
        assert (out[::8, ::8].reshape(-1) >> 24).tolist() == [r["cls"] for r in rows]
    for g in kat["flood_fill"]:
        cl = np.array(g["classes"])
        cells = np.full((1, 784, 81), -1.0, np.float32)
        for k in (1, 2, 3):
            cells[0, cl == k, k] = 1.0
        if g["diverges"]:
            with pytest.raises(ya.YhError) as e:
                eng.postprocess_cells(cells, ya.COMPAT_STRICT)
            assert e.value.code == ya.EDIVERGE
        else:
            out = eng.postprocess_cells(cells, ya.COMPAT_STRICT)[0]
            assert np.array_equal(out[::8, ::8].reshape(-1), (cl.astype(np.uint32) << 24))   # ids all -1: cls<<24


@pytest.mark.parametrize("mode", [0, 1])
def test_postprocess_random_vs_oracle(eng, oracle, mode):
    import yolact_amd as ya
    rng = np.random.default_rng(mode)
    for trial in range(6):
        cells = rng.normal(-0.5 if mode == 0 else 0.0, 1.0, (2, 784, 81)).astype(np.float32)
        if mode == 0:
            cells[:, :, 3] = -1.0            # no balls: the strict path terminates
            cells[:, rng.integers(0, 784, 3), 3] = 5.0  # a few isolated balls (may or may not touch)
        want = [oracle.postprocess_tile(cells[t], 28, 81, mode) for t in range(2)]
        if any(rc for rc, _ in want):
            with pytest.raises(ya.YhError):
                eng.postprocess_cells(cells, mode)
            continue
        got = eng.postprocess_cells(cells, mode)
        for t in range(2):
            assert np.array_equal(got[t].reshape(-1), want[t][1])


def test_classify_frame_end_to_end(eng, oracle, golden_dir):
    """Yolact::classify (yolact.rs:192-234) in place on a 640x480 packed frame. The network sits in
    the middle, so the oracle's post half is fed the engine's own output 4: everything around the
    network must then match bit for bit."""
    from PIL import Image
    import yolact_amd as ya
    rgb = np.asarray(Image.open(os.path.join(golden_dir, "frc_balls.png")).convert("RGB").resize((640, 480), Image.BILINEAR))
    frame0 = oracle.pack_rgb(rgb)
    tiles = oracle.classify_pre(frame0, 640, 480, 224)
    frame = frame0.copy()
    try:
        eng.classify_frame(frame, 640, 480, ya.COMPAT_STRICT)
        mode = 0
    except ya.YhError as e:
        assert e.code == ya.EDIVERGE
        assert np.array_equal(frame, frame0)          # untouched on divergence
        eng.classify_frame(frame, 640, 480, ya.COMPAT_SANE)
        mode = 1
    cells = eng.output(4)                              # the two tiles that classify just ran
    assert cells.shape == (2, 28, 28, 81)
    # pre-processing half: run the same tiles through invoke and compare output 4
    eng.set_input(tiles)
    eng.invoke()
    assert np.array_equal(eng.output(4), cells)
    rc, want = oracle.classify_post(cells, 224, 81, mode, 640, 480)
    assert rc == 0 and np.array_equal(frame, want)
    if mode == 0:
        assert (oracle.consumer_low16(frame) == 0).all()   # A10: what scene.rs:93 keeps


def test_yolact_mirror_surface(built, oracle, golden_dir):
    """`Yolact.init()` / `classify(&mut [u32])` — the reference's two public items."""
    from PIL import Image
    import yolact_amd as ya
    y = ya.Yolact.init(seed=1, compat_mode=ya.COMPAT_SANE)
    rgb = np.asarray(Image.open(os.path.join(golden_dir, "red_robot.png")).convert("RGB").resize((640, 480), Image.BILINEAR))
    buf = oracle.pack_rgb(rgb)
    before = buf.copy()
    y.classify(buf)
    assert buf.shape == before.shape and not np.array_equal(buf, before)
    assert ((buf >> 24) <= 3).all() and (buf & 0xFFFF == 0).all()
    y.interpreter.close()
