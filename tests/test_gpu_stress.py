"""-m gpu: seeded random sweeps over shapes x tile variants on small-integer data (every product and partial sum
exact, so any summation order gives the same bits): the conv kernel's tile variants and forced split-K, the
multi-level form, the fused stem + pool - each against the oracle bit for bit. A longer run of the same
generators (1 182 + 263 + 340 cases) was clean when they were written."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TILES = {0: "128x128", 1: "64x256", 2: "32x256", 5: "128x256", 7: "128x128_S3", 8: "256x256_M16", 12: "128x128_M16",
         13: "128x128_S3_M16", 15: "128x256_M16", 16: "64x64_S3", 21: "128x128_K1", 22: "64x256_K1"}


@pytest.fixture(scope="module")
def eng(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=64, max_batch=1, use_graph=False)
    yield e
    e.close()


def _forced(eng, tune, fn):
    """Runs fn with the single-op test knobs (op_tile, op_kslices) of THIS handle set, then clears them."""
    eng.set_tuning(**tune)
    try:
        return fn()
    finally:
        eng.reset_tuning("op_tile", "op_kslices")


def test_conv_random_shapes_and_tiles(eng, oracle):
    rng = np.random.default_rng(2024)
    done = 0
    while done < 160:
        k = int(rng.choice([1, 3, 3, 5])); stride = int(rng.choice([1, 1, 2])); pad = int(rng.choice([0, k // 2]))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(k, 24)), int(rng.integers(k, 24))
        cin = int(rng.choice([64, 128, 192])); cout = int(rng.choice([8, 24, 64, 72, 128, 200, 256, 351, 512]))
        tile = int(rng.choice(list(TILES)))
        if (tile in (1, 22) and cout > 64) or (tile == 2 and cout > 32):
            tile = 0
        res, act = bool(rng.integers(0, 2)), int(rng.integers(0, 2))
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        if ho < 1 or wo < 1:
            continue
        env = {"op_tile": tile}
        ksl = int(rng.choice([0, 0, 2, 3])) if tile in (7, 16) else 0
        if ksl and k * k * cin // 64 >= ksl:
            env["op_kslices"] = ksl
        x = rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)
        wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
        b = rng.integers(-4, 5, cout).astype(np.float32)
        r = rng.integers(-5, 6, (n, ho, wo, cout)).astype(np.float32) if res else None
        y = _forced(eng, env, lambda: eng.op_conv2d(x, wt, b, stride, pad, r, act))
        assert np.array_equal(y, oracle.conv2d(x, wt, b, stride, pad, r, act, f16=True)), (TILES[tile], n, h, w, cin, cout, k, stride, pad, res, act, env)
        done += 1


def test_multilevel_random_levels_and_tiles(eng, oracle):
    rng = np.random.default_rng(7)
    for _ in range(50):
        sizes = [int(rng.integers(1, 14)) for _ in range(int(rng.integers(1, 6)))]
        n, cin, cout, k = int(rng.integers(1, 4)), int(rng.choice([64, 128])), int(rng.choice([64, 128, 256, 351])), int(rng.choice([1, 3]))
        tile = int(rng.choice([0, 5, 7, 8, 12, 13, 15, 16]))
        env = {"op_tile": tile}
        ksl = int(rng.choice([0, 2, 3])) if tile in (7, 16) else 0
        if ksl and k * k * cin // 64 >= ksl:
            env["op_kslices"] = ksl
        cells = sum(s * s for s in sizes)
        x = rng.integers(-3, 4, (n, cells, cin)).astype(np.float32)
        wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
        b = rng.integers(-4, 5, cout).astype(np.float32)
        y = _forced(eng, env, lambda: eng.op_conv2d_levels(x, sizes, wt, b, act=1))
        off = 0
        for s_ in sizes:
            yo = oracle.conv2d(x[:, off:off + s_ * s_].reshape(n, s_, s_, cin), wt, b, 1, k // 2, None, 1, f16=True)
            assert np.array_equal(y[:, off:off + s_ * s_], yo.reshape(n, s_ * s_, cout)), (TILES[tile], sizes, n, cin, cout, k, env)
            off += s_ * s_


def test_stem_pool_random_sizes(eng, oracle):
    rng = np.random.default_rng(11)
    for _ in range(40):
        n, S = int(rng.integers(1, 4)), 2 * int(rng.integers(4, 60))
        x = rng.integers(-4, 5, (n, S, S, 3)).astype(np.float32)
        wt = rng.integers(-3, 4, (64, 7, 7, 3)).astype(np.float32)
        b = rng.integers(-30, 31, 64).astype(np.float32)
        stem, pool = eng.op_stem_pool(x, wt, b)
        so = oracle.conv2d(x, wt, b, 2, 3, None, 1, f16=True)
        assert np.array_equal(stem, so) and np.array_equal(pool, oracle.maxpool3x3s2(so)), (n, S)


@pytest.mark.parametrize("S,C,top_k,max_dets", [(96, 81, 8, 5), (96, 21, 50, 100), (320, 81, 200, 100), (320, 81, 37, 11)])
def test_tail_random_heads_bit_exact(built, oracle, S, C, top_k, max_dets):
    """Seeded random head tensors through yh_op_detect against the oracle, bit for bit: candidate counts from
    zero to every prior of a class (more than the 4 096 keys the class kernel stages in LDS at S = 320:
    the unstaged radix-select path), exact score ties, top_k and max_dets that are not powers of two."""
    import yolact_amd as ya
    eng = ya.Engine(input_size=S, max_batch=2, use_graph=False, num_classes=C, top_k=top_k, max_dets=max_dets)
    P, hp = eng.P, eng.hp
    pri = eng.priors()
    rng = np.random.default_rng(S + C + top_k)
    h = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    for trial in range(3):
        loc = h(rng.normal(0, 0.5, (2, P, 4)))
        conf = rng.normal(0, 1.5, (2, P, C))
        conf[:, :, 0] += 4.0
        hot = int(rng.integers(1, C))
        if trial == 0:
            conf[0, :, hot] = 9.0                                   # every prior a candidate of one class, all tied
        elif trial == 1:
            conf[0, :, hot] += rng.normal(6.0, 0.3, P)              # every prior a candidate, distinct scores
            conf[1] = -5.0; conf[1, :, 0] = 5.0                     # frame 1: no candidate at all
        conf = h(conf)
        mask = h(np.tanh(rng.normal(0, 1, (2, P, 32))))
        proto = h(np.maximum(rng.normal(0, 1, (2, hp, hp, 32)), 0))
        eng.op_detect(loc, conf, mask, proto)
        for f in range(2):
            d, m = eng.detections(f)
            od, om = oracle.detect(loc[f], conf[f], mask[f], proto[f], pri, num_classes=C, top_k=top_k, max_dets=max_dets)
            assert [(x["class_id"], x["prior"], x["score"], x["box"]) for x in d] == [(x["class_id"], x["prior"], x["score"], x["box"]) for x in od], (trial, f)
            assert np.array_equal(m, om), (trial, f)
    eng.close()
