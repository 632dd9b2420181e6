"""-m gpu: seeded random sweeps over shapes x tile variants on small-integer data (every product and partial sum
exact, so any summation order gives the same bits): the conv kernel's tile variants and forced split-K, the
multi-level form, the fused stem + pool - each against the oracle bit for bit. A longer run of the same
generators (1 182 + 263 + 340 cases) was clean when they were written."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TILES = {0: "128x128", 1: "64x256", 2: "32x256", 5: "128x256", 7: "128x128_S3", 8: "256x256_M16", 12: "128x128_M16",
         13: "128x128_S3_M16", 15: "128x256_M16", 16: "64x64_S3"}


@pytest.fixture(scope="module")
def eng(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=64, max_batch=1, use_graph=False)
    yield e
    e.close()


def _with_env(env, fn):
    keys = ("YH_OP_TILE", "YH_OP_KSLICES")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


def test_conv_random_shapes_and_tiles(eng, oracle):
    rng = np.random.default_rng(2024)
    done = 0
    while done < 160:
        k = int(rng.choice([1, 3, 3, 5])); stride = int(rng.choice([1, 1, 2])); pad = int(rng.choice([0, k // 2]))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(k, 24)), int(rng.integers(k, 24))
        cin = int(rng.choice([64, 128, 192])); cout = int(rng.choice([8, 24, 64, 72, 128, 200, 256, 351, 512]))
        tile = int(rng.choice(list(TILES)))
        if (tile == 1 and cout > 64) or (tile == 2 and cout > 32):
            tile = 0
        res, act = bool(rng.integers(0, 2)), int(rng.integers(0, 2))
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        if ho < 1 or wo < 1:
            continue
        env = {"YH_OP_TILE": tile}
        ksl = int(rng.choice([0, 0, 2, 3])) if tile in (7, 16) else 0
        if ksl and k * k * cin // 64 >= ksl:
            env["YH_OP_KSLICES"] = ksl
        x = rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)
        wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
        b = rng.integers(-4, 5, cout).astype(np.float32)
        r = rng.integers(-5, 6, (n, ho, wo, cout)).astype(np.float32) if res else None
        y = _with_env(env, lambda: eng.op_conv2d(x, wt, b, stride, pad, r, act))
        assert np.array_equal(y, oracle.conv2d(x, wt, b, stride, pad, r, act, f16=True)), (TILES[tile], n, h, w, cin, cout, k, stride, pad, res, act, env)
        done += 1


def test_multilevel_random_levels_and_tiles(eng, oracle):
    rng = np.random.default_rng(7)
    for _ in range(50):
        sizes = [int(rng.integers(1, 14)) for _ in range(int(rng.integers(1, 6)))]
        n, cin, cout, k = int(rng.integers(1, 4)), int(rng.choice([64, 128])), int(rng.choice([64, 128, 256, 351])), int(rng.choice([1, 3]))
        tile = int(rng.choice([0, 5, 7, 8, 12, 13, 15, 16]))
        env = {"YH_OP_TILE": tile}
        ksl = int(rng.choice([0, 2, 3])) if tile in (7, 16) else 0
        if ksl and k * k * cin // 64 >= ksl:
            env["YH_OP_KSLICES"] = ksl
        cells = sum(s * s for s in sizes)
        x = rng.integers(-3, 4, (n, cells, cin)).astype(np.float32)
        wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
        b = rng.integers(-4, 5, cout).astype(np.float32)
        y = _with_env(env, lambda: eng.op_conv2d_levels(x, sizes, wt, b, act=1))
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
