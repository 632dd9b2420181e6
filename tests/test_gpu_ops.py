"""-m gpu: single HIP kernels through the C ABI vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def f16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def eng(built):
    import yolact_amd as ya
    e = ya.Engine(input_size=128, max_batch=2, use_graph=False)
    yield e
    e.close()


def test_mfma_operand_and_output_maps(eng, oracle):
    """A = I with an ASYMMETRIC B: a 1x1 conv whose weight is the identity must copy x exactly,
    and a weight with one 1 per row at a shifted column must permute channels: catches swapped
    row/col or k-order maps of v_mfma_f32_32x32x16_f16."""
    rng = np.random.default_rng(1)
    c = 128
    x = f16(rng.integers(-8, 9, (1, 5, 7, c)))
    w = np.zeros((c, 1, 1, c), np.float32)
    w[np.arange(c), 0, 0, (np.arange(c) * 37 + 11) % c] = 1.0
    y = eng.op_conv2d(x, w, np.zeros(c, np.float32))
    assert np.array_equal(y, x[..., (np.arange(c) * 37 + 11) % c])


CASES = [  # n, h, w, cin, cout, k, stride, pad, res, act
    (1, 8, 8, 64, 128, 1, 1, 0, False, 0),
    (1, 9, 7, 64, 64, 3, 1, 1, False, 1),
    (2, 17, 13, 128, 256, 3, 2, 1, True, 1),
    (1, 20, 20, 256, 32, 1, 1, 0, False, 1),
    (1, 12, 12, 256, 351, 3, 1, 1, False, 2),
    (1, 33, 31, 3, 64, 7, 2, 3, False, 1),
    (2, 16, 16, 256, 256, 3, 1, 1, False, 1),
    (1, 5, 5, 256, 256, 3, 2, 1, False, 0),      # FPN downsample on a 5x5 map
    (3, 1, 1, 256, 351, 3, 1, 1, False, 2),      # 1x1 spatial, all taps but the centre padded
    (1, 35, 35, 512, 128, 1, 2, 0, False, 0),    # strided 1x1 projection
    (1, 18, 18, 2048, 256, 1, 1, 0, True, 0),    # deep K
    (1, 40, 40, 64, 12, 3, 1, 1, False, 0),      # tiny cout
]


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res,act", CASES)
def test_conv_vs_oracle(eng, oracle, n, h, w, cin, cout, k, stride, pad, res, act):
    rng = np.random.default_rng(n * 7 + cin + cout)
    x = f16(rng.normal(0, 1, (n, h, w, cin)))
    wt = f16(rng.normal(0, 1, (cout, k, k, cin)) / np.sqrt(k * k * cin))
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = f16(rng.normal(0, 1, (n, ho, wo, cout))) if res else None
    y = eng.op_conv2d(x, wt, b, stride, pad, r, act)
    yo = oracle.conv2d(x, wt, b, stride, pad, r, act, f16=True)
    # f32 accumulate on both sides; only summation order differs -> at most 1 f16 ulp apart
    tol = 2.0 ** -10 * np.maximum(np.abs(yo), 1.0) + 1e-3
    assert (np.abs(y - yo) <= tol).all(), float(np.abs(y - yo).max())


def test_conv_exact_on_integers(eng, oracle):
    """Small-integer data: every product and partial sum is exact in f32, so any summation order
    gives the same bits; the kernel must match the oracle bit for bit (indexing, padding, bias)."""
    rng = np.random.default_rng(9)
    x = rng.integers(-3, 4, (2, 11, 9, 64)).astype(np.float32)
    wt = rng.integers(-2, 3, (128, 3, 3, 64)).astype(np.float32)
    b = rng.integers(-4, 5, 128).astype(np.float32)
    r = rng.integers(-5, 6, (2, 11, 9, 128)).astype(np.float32)
    y = eng.op_conv2d(x, wt, b, 1, 1, r, 1)
    yo = oracle.conv2d(x, wt, b, 1, 1, r, 1, f16=True)
    assert np.array_equal(y, yo)


@pytest.mark.parametrize("cin,cout,k", [(64, 256, 3), (256, 512, 1), (128, 128, 3), (64, 351, 3)])
def test_conv_exact_on_integers_big_tiles(eng, oracle, cin, cout, k):
    """Same exactness check on the 8-wave tiles (256x256 with the 16x16x32 MFMA shape and the split
    epilogue; 128x256 on the 3-stage ring), ragged M and channel tails included."""
    rng = np.random.default_rng(cin + cout)
    x = rng.integers(-3, 4, (3, 13, 11, cin)).astype(np.float32)
    wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    y = eng.op_conv2d(x, wt, b, 1, k // 2, None, 0)
    yo = oracle.conv2d(x, wt, b, 1, k // 2, None, 0, f16=True)
    assert np.array_equal(y, yo)


@pytest.mark.parametrize("cin,cout,k,n,h,w", [
    (64, 256, 3, 5, 17, 16),     # 6 row tiles on "4 CUs": one whole round on the big tile + a tail on 128x128 tiles
    (512, 512, 1, 5, 16, 16),    # two channel tiles x 5 row tiles: 2 rounds + a tail of 2 workgroups
    (64, 351, 3, 5, 17, 12),     # 384 padded channels, 4 row tiles: 256-wide launch + 128-wide launch
    (512, 351, 1, 3, 21, 16),
])
def test_conv_split_launch_plans_exact_on_integers(eng, oracle, cin, cout, k, n, h, w):
    """Two-phase (wave-quantisation tail) and channel-split launch plans, reached with small
    tensors by planning for a 4-CU chip (tune.plan_cus = 4 on this handle): same bits as the oracle, and the same
    bits as the single-launch plan."""
    rng = np.random.default_rng(cin + cout + k)
    x = f16(rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32))
    wt = f16(rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32))
    b = rng.integers(-4, 5, cout).astype(np.float32)
    r = f16(rng.integers(-5, 6, (n, h, w, cout)).astype(np.float32))
    single = eng.op_conv2d(x, wt, b, 1, k // 2, r, 1)
    assert eng.last_conv_launches() == 1
    eng.set_tuning(plan_cus=4)
    try:
        split = eng.op_conv2d(x, wt, b, 1, k // 2, r, 1)
        assert eng.last_conv_launches() == 2          # the plan under test was really taken
    finally:
        eng.reset_tuning("plan_cus")
    yo = oracle.conv2d(x, wt, b, 1, k // 2, r, 1, f16=True)
    assert np.array_equal(split, yo) and np.array_equal(single, yo)


@pytest.mark.parametrize("k", [1, 3])
def test_head_remainder_on_96_channel_tiles_exact_on_integers(eng, oracle, k):
    """Round 5: the shared head's 351 channels run as a 256-wide launch plus a launch for channels 256 .. 350 - 95 of them - which in its
    multi-level streaming form takes 96-channel tiles (conv_igemm_f16<96,128,...,mfma16>, a channel base that is no multiple of
    the tile) instead of 128-channel ones. Reached with small tensors by planning for a 4-CU chip; every level equal to the
    oracle's convolution of that level alone, bit for bit (channel 351 of the padded row must stay untouched: the op's rows are
    352 wide and are compared whole), and equal to the 128-channel form (tune.chsplit = 2)."""
    rng = np.random.default_rng(96 + k)
    sizes, n, cin, cout = [13, 9, 5, 3], 6, 64 if k == 3 else 512, 351       # 284 cells x 6 = 1704 rows: 7 row tiles of 256
    cells = sum(s * s for s in sizes)
    x = rng.integers(-3, 4, (n, cells, cin)).astype(np.float32)
    wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    outs = {}
    for chsplit in (1, 2):
        eng.set_tuning(plan_cus=4, chsplit=chsplit)
        try:
            outs[chsplit] = eng.op_conv2d_levels(x, sizes, wt, b, act=1)
            assert eng.last_conv_launches() == 2
        finally:
            eng.reset_tuning("plan_cus", "chsplit")
    off = 0
    for s_ in sizes:
        xl = x[:, off:off + s_ * s_].reshape(n, s_, s_, cin)
        yo = oracle.conv2d(f16(xl), f16(wt), b, 1, k // 2, None, 1, f16=True).reshape(n, s_ * s_, cout)
        assert np.array_equal(outs[1][:, off:off + s_ * s_], yo), s_
        off += s_ * s_
    assert np.array_equal(outs[1], outs[2])


def test_conv_two_phase_plan_is_bitwise_identical_on_random_data(eng):
    """Random (non-exact) data: the tail phase's 128x128 16x16x32 tiles must accumulate each output
    element in the same order as the 256x256 tile, so the plan cannot change a single bit."""
    rng = np.random.default_rng(77)
    x = f16(rng.normal(0, 1, (5, 17, 16, 128)))
    wt = f16(rng.normal(0, 1, (256, 3, 3, 128)) / 34)
    b = rng.normal(0, 0.1, 256).astype(np.float32)
    single = eng.op_conv2d(x, wt, b, 1, 1, None, 1)
    eng.set_tuning(plan_cus=4)
    try:
        split = eng.op_conv2d(x, wt, b, 1, 1, None, 1)
        assert eng.last_conv_launches() == 2
    finally:
        eng.reset_tuning("plan_cus")
    assert np.array_equal(single, split)


@pytest.mark.parametrize("n,S", [(1, 16), (2, 30), (1, 64), (3, 34), (1, 96)])
def test_stem_pool_fused_exact_on_integers(eng, oracle, n, S):
    """Fused 7x7/2 stem + 3x3/2 max pool on small integers (every partial sum exact): the pre-pool
    tensor it can emit and the pooled tensor must equal the oracle's conv2d -> maxpool bit for bit.
    Sizes cover one tile, ragged tile edges (PO % 8 != 0), several images, and the image borders
    where stem pixels and pool taps fall outside."""
    rng = np.random.default_rng(S)
    x = rng.integers(-4, 5, (n, S, S, 3)).astype(np.float32)
    wt = rng.integers(-3, 4, (64, 7, 7, 3)).astype(np.float32)
    b = rng.integers(-30, 31, 64).astype(np.float32)
    stem, pool = eng.op_stem_pool(x, wt, b)
    so = oracle.conv2d(x, wt, b, 2, 3, None, 1, f16=True)
    assert np.array_equal(stem, so)
    assert np.array_equal(pool, oracle.maxpool3x3s2(so))
    _, pool2 = eng.op_stem_pool(x, wt, b, want_stem=False)     # production form: no stem output
    assert np.array_equal(pool2, pool)


@pytest.mark.parametrize("n,S", [(2, 30), (1, 64), (1, 98)])
def test_stem_pool_with_fused_preprocessing_is_bitwise_the_two_step_path(eng, n, S):
    """Production runs hand the fused stem raw uint8 frames (its loader normalises them); that must equal,
    bit for bit, normalising first ((v - mean) / std in f32, rounded to f16: the preprocess kernel's
    expression) and running the f16 form - image borders included, where the padding is exact zero
    and NOT a normalised black pixel."""
    rng = np.random.default_rng(S)
    rgb = rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8)
    wt = f16(rng.normal(0, 1, (64, 7, 7, 3)) / 12)
    b = rng.normal(0, 0.2, 64).astype(np.float32)
    mean = np.array([123.68, 116.78, 103.94], np.float32); sd = np.array([58.40, 57.12, 57.38], np.float32)
    x = ((rgb.astype(np.float32) - mean) / sd).astype(np.float16).astype(np.float32)
    stem_a, pool_a = eng.op_stem_pool_rgb8(rgb, wt, b)
    stem_b, pool_b = eng.op_stem_pool(x, wt, b)
    assert np.array_equal(stem_a, stem_b) and np.array_equal(pool_a, pool_b)
    assert np.abs(stem_a).max() > 0.5


def test_stem_pool_fused_vs_oracle_random(eng, oracle):
    rng = np.random.default_rng(5)
    x = f16(rng.normal(0, 1.2, (2, 70, 70, 3)))
    wt = f16(rng.normal(0, 1, (64, 7, 7, 3)) / 12)
    b = rng.normal(0, 0.2, 64).astype(np.float32)
    stem, pool = eng.op_stem_pool(x, wt, b)
    so = oracle.conv2d(x, wt, b, 2, 3, None, 1, f16=True)
    assert np.abs(stem - so).max() <= 2.0 ** -10 * max(1.0, np.abs(so).max()) + 1e-3
    assert np.array_equal(pool, oracle.maxpool3x3s2(stem))       # the pool of ITS stem is exact


# ConvTile ids (csrc/yh_internal.h): the variants a head conv can be launched on
_TILES = {"128x128": 0, "128x256": 5, "128x128_S3": 7, "256x256_M16": 8, "128x128_M16": 12, "128x128_S3_M16": 13, "128x256_M16": 15, "64x64_S3": 16}


def _forced(eng, tune, fn):
    eng.set_tuning(**tune)
    try:
        return fn()
    finally:
        eng.reset_tuning(*tune)


@pytest.mark.parametrize("tile,kslices", [(t, 0) for t in _TILES] + [("128x128_S3", 3), ("64x64_S3", 4)])
@pytest.mark.parametrize("k", [3, 1])
def test_conv_multilevel_exact_on_integers(eng, oracle, tile, kslices, k):
    """The shared head's multi-level form (one launch over pyramid levels laid end to end) on every tile
    variant it can be launched on, split-K included: each level must equal the oracle's convolution of
    that level alone, bit for bit - in particular no tap may reach into the neighbouring level's cells
    (level edges 9, 5, 3, 1: every level boundary falls inside a tile)."""
    rng = np.random.default_rng(k)
    sizes, n, cin, cout = [9, 5, 3, 1], 3, 128, 256
    cells = sum(s * s for s in sizes)
    x = rng.integers(-3, 4, (n, cells, cin)).astype(np.float32)
    wt = rng.integers(-2, 3, (cout, k, k, cin)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    env = {"op_tile": _TILES[tile]}
    if kslices: env.update(op_kslices=kslices)
    y = _forced(eng, env, lambda: eng.op_conv2d_levels(x, sizes, wt, b, act=1))
    off = 0
    for s_ in sizes:
        xl = x[:, off:off + s_ * s_].reshape(n, s_, s_, cin)
        yo = oracle.conv2d(f16(xl), f16(wt), b, 1, k // 2, None, 1, f16=True).reshape(n, s_ * s_, cout)
        assert np.array_equal(y[:, off:off + s_ * s_], yo), (tile, s_)
        off += s_ * s_


@pytest.mark.parametrize("tile,kslices", [("128x128_S3", 0), ("64x64_S3", 0), ("128x128_S3", 2), ("64x64_S3", 5), ("128x128_M16", 0), ("128x128_S3_M16", 0)])
def test_conv_small_tile_variants_exact_on_integers(eng, oracle, tile, kslices):
    """The latency-bound tile variants the engine picks at small batch (3-stage 128x128 and 64x64 rings,
    their split-K forms, the 16x16x32 tail tiles), forced through the op entry: ragged M, residual, ReLU."""
    rng = np.random.default_rng(3)
    x = rng.integers(-3, 4, (2, 13, 11, 128)).astype(np.float32)
    wt = rng.integers(-2, 3, (192, 3, 3, 128)).astype(np.float32)
    b = rng.integers(-4, 5, 192).astype(np.float32)
    r = rng.integers(-5, 6, (2, 13, 11, 192)).astype(np.float32)
    env = {"op_tile": _TILES[tile]}
    if kslices: env.update(op_kslices=kslices)
    y = _forced(eng, env, lambda: eng.op_conv2d(f16(x), f16(wt), b, 1, 1, f16(r), 1))
    assert np.array_equal(y, oracle.conv2d(f16(x), f16(wt), b, 1, 1, f16(r), 1, f16=True))


_DUAL_TILES = {"128x128": 0, "128x128_K1": 21, "128x128_S3": 7, "64x64_S3": 16, "128x128_M16": 12, "128x128_S3_M16": 13, "256x256_M16": 8}


@pytest.mark.parametrize("tile,kslices", [(t, 0) for t in _DUAL_TILES] + [("128x128_S3", 2), ("128x128_S3", 3), ("64x64_S3", 2), ("64x64_S3", 5)])
@pytest.mark.parametrize("stride2,c1,c2", [(1, 64, 64), (2, 128, 256), (2, 64, 192)])
def test_dual_source_conv_exact_on_integers(eng, oracle, tile, kslices, stride2, c1, c2):
    """The two-source 1x1 form (a bottleneck block's last conv + its projection shortcut as one accumulation over
    [x1 | x2 at stride2]) on every tile it is instantiated for, split-K slices starting before, at and after the switch
    of sources: equal, bit for bit, to the oracle's 1x1 conv of the channel-concatenated tensors. Ragged M (two images
    straddling tiles), odd source size under stride 2 (the last source row / column is never read)."""
    rng = np.random.default_rng(c1 + c2 + stride2)
    n, ho, wo, cout = 2, 13, 11, 256
    h2, w2 = (ho - 1) * stride2 + 1 + (stride2 - 1), (wo - 1) * stride2 + 1
    x1 = rng.integers(-3, 4, (n, ho, wo, c1)).astype(np.float32)
    x2 = rng.integers(-3, 4, (n, h2, w2, c2)).astype(np.float32)
    wt = rng.integers(-2, 3, (cout, c1 + c2)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    env = {"op_tile": _DUAL_TILES[tile]}
    if kslices: env.update(op_kslices=kslices)
    y = _forced(eng, env, lambda: eng.op_conv2d_dual(x1, x2, stride2, wt, b, act=1))
    cat = np.concatenate([x1, x2[:, ::stride2, ::stride2][:, :ho, :wo]], axis=-1)
    yo = oracle.conv2d(f16(cat), f16(wt.reshape(cout, 1, 1, c1 + c2)), b, 1, 0, None, 1, f16=True)
    assert np.array_equal(y, yo), tile


def test_dual_source_conv_vs_oracle_on_reals(eng, oracle):
    """... and on real-valued data (the engine's shapes of layer 2, block 0, on a small map): within one f16 ulp of the
    oracle's conv of the concatenated tensors (f32 accumulation on both sides, different summation order)."""
    rng = np.random.default_rng(11)
    n, ho, wo, c1, c2, cout = 2, 9, 9, 128, 256, 512
    x1 = f16(rng.normal(0, 1, (n, ho, wo, c1)))
    x2 = f16(rng.normal(0, 1, (n, 2 * ho, 2 * wo, c2)))
    wt = f16(rng.normal(0, 1, (cout, c1 + c2)) / np.sqrt(c1 + c2))
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    y = eng.op_conv2d_dual(x1, x2, 2, wt, b, act=1)
    cat = np.concatenate([x1, x2[:, ::2, ::2]], axis=-1)
    yo = oracle.conv2d(cat, wt.reshape(cout, 1, 1, c1 + c2), b, 1, 0, None, 1, f16=True)
    assert np.all(np.abs(y - yo) <= 2.0 ** -10 * np.maximum(np.abs(yo), 1.0) + 1e-3)


@pytest.mark.parametrize("h,w,ho,wo", [(18, 18, 35, 35), (35, 35, 69, 69), (4, 4, 8, 8), (5, 7, 9, 13),
                                       (69, 69, 138, 138), (6, 10, 12, 20), (9, 5, 18, 10), (2, 2, 4, 4), (1, 1, 2, 2), (3, 1, 6, 2)])
def test_bilinear_bit_exact(eng, oracle, h, w, ho, wo):
    """(the exact x2 cases run bilinear2x_f16: four source pixels shared by 4 x 2 outputs of a lane, borders on the generic path)"""
    x = f16(np.random.default_rng(h).normal(0, 2, (2, h, w, 64)))
    assert np.array_equal(eng.op_bilinear(x, ho, wo), oracle.bilinear(x, ho, wo, f16=True))


@pytest.mark.parametrize("h,w", [(275, 9), (12, 12), (7, 9), (1, 1)])
def test_maxpool_bit_exact(eng, oracle, h, w):
    x = f16(np.random.default_rng(w).normal(0, 2, (2, h, w, 64)))
    assert np.array_equal(eng.op_maxpool(x), oracle.maxpool3x3s2(x))


@pytest.mark.parametrize("inv_scale", [1.0, 0.37, 8.0, 1.0 / 448.0])
def test_quantize_e4m3_every_f16_bit_exact(eng, oracle, inv_scale):
    """f16 -> OCP FP8 E4M3 on the device (hardware conversion behind a clamp) against the oracle's encoder
    for all 65 536 f16 bit patterns: nearest-even rounding, saturation at +-448, subnormals, zeros, NaN."""
    bits = np.arange(65536, dtype=np.uint16)
    got = eng.op_quantize_e4m3(bits, inv_scale)
    x = bits.view(np.float16).astype(np.float32)
    want = oracle.quantize_e4m3(x, inv_scale)
    nan = np.isnan(x)
    assert np.array_equal(got[~nan], want[~nan])
    assert np.all((got[nan] & 0x7F) == 0x7F)


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res", [
    (2, 13, 11, 128, 256, 3, 1, 1, True),      # ragged M, images straddling tiles, borders
    (1, 16, 16, 256, 512, 1, 1, 0, False),     # two channel tiles, two K steps
    (3, 9, 9, 128, 200, 3, 2, 1, False),       # stride 2, channel tail (cout not a multiple of 8 * 32)
])
def test_conv_fp8_exact_on_integers(eng, oracle, n, h, w, cin, cout, k, stride, pad, res):
    """Experimental fp8 convolution (E4M3 operands on the block-scaled MFMA, f32 accumulate): with
    small-integer operands every product and sum is exact, so the result must equal the oracle's
    convolution of the DECODED operands, scaled per channel (powers of two), bit for bit."""
    rng = np.random.default_rng(cin + cout + k)
    xi = rng.integers(-4, 5, (n, h, w, cin)).astype(np.float32)
    wi = rng.integers(-3, 4, (cout, k, k, cin)).astype(np.float32)
    xc, wc = oracle.quantize_e4m3(xi), oracle.quantize_e4m3(wi)
    table = oracle.e4m3_decode_table()
    assert np.array_equal(table[xc], xi) and np.array_equal(table[wc], wi)          # the codes carry the integers exactly
    scale = (2.0 ** rng.integers(-3, 1, cout)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = rng.integers(-5, 6, (n, ho, wo, cout)).astype(np.float32) if res else None
    y, _ = eng.op_conv2d_fp8(xc, wc, scale, b, stride, pad, r, 1)
    acc = oracle.conv2d(xi, wi, np.zeros(cout, np.float32), stride, pad, None, 0, f16=False)
    want = acc * scale + b
    if res: want = want + r
    want = np.maximum(want, 0).astype(np.float16).astype(np.float32)
    assert np.array_equal(y, want)
