"""-m gpu: error behaviour of the C ABI (the reference `.expect()`s; the ABI returns a code and a
message and must never crash, truncate silently or fall back)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_rejects_bad_configs(built):
    import yolact_amd as ya
    for kw in (dict(input_size=32), dict(input_size=4096), dict(max_batch=0), dict(max_batch=1000), dict(num_classes=3),
               dict(top_k=0), dict(top_k=1000), dict(max_dets=0), dict(max_dets=500), dict(backbone=34), dict(device=99)):
        with pytest.raises(ya.YhError):
            ya.Engine(**{**dict(input_size=128), **kw})
    L = ya.load_library()
    cfg = ya.Config()
    L.yh_default_config(C.byref(cfg))
    cfg.abi_version = 999
    h = C.c_void_p()
    assert L.yh_create(C.byref(cfg), C.byref(h)) == -1 and b"ABI" in L.yh_last_error(None)


def test_call_order_and_bounds(built):
    import yolact_amd as ya
    eng = ya.Engine(input_size=128, max_batch=2, use_graph=False)
    L = eng.L
    with pytest.raises(ya.YhError) as e:
        eng.evaluate()
    assert e.value.code == -5                                   # no weights
    eng.load_weights(eng.generate_weights(1))
    with pytest.raises(ya.YhError):
        eng.evaluate()                                          # no input yet
    with pytest.raises(ya.YhError):
        eng.output(0)
    x = np.zeros((3, 128, 128, 3), np.uint8)
    assert L.yh_set_input_u8(eng.h, x.ctypes.data_as(C.c_void_p), 3) == -1    # n > max_batch
    assert L.yh_set_input_u8(eng.h, x.ctypes.data_as(C.c_void_p), 0) == -1    # empty batch
    eng.set_input(x[:1])
    eng.evaluate()
    nd = C.c_int32()
    assert L.yh_read_detections(eng.h, 1, C.byref(nd), None, 0, None, 0) == -1     # frame >= n
    assert L.yh_read_detections(eng.h, 0, C.byref(nd), None, 0, None, 0) == 0 and nd.value >= 0
    if nd.value:
        dets = (ya.Detection * 1)()
        small = np.zeros(4, np.uint8)
        assert L.yh_read_detections(eng.h, 0, C.byref(nd), C.cast(dets, C.c_void_p), 0, None, 0) == -1            # capacity too small
        assert L.yh_read_detections(eng.h, 0, C.byref(nd), None, 0, small.ctypes.data_as(C.c_void_p), 4) == -1    # mask capacity
    out = np.zeros(8, np.float32)
    assert L.yh_output_read_f32(eng.h, 0, out.ctypes.data_as(C.c_void_p), 8) == -1
    assert L.yh_output_read_f32(eng.h, 7, out.ctypes.data_as(C.c_void_p), 8) == -1
    info = ya.capi.TensorInfo()
    assert L.yh_output_info(eng.h, 5, C.byref(info)) == -1
    d4 = (C.c_int32 * 4)()
    assert L.yh_debug_read_tensor(eng.h, b"no_such_layer", None, 0, C.byref(d4)) == -1
    assert b"no_such_layer" in L.yh_last_error(eng.h)
    # the pre-pool stem tensor is fused away unless the engine was created with debug_tensors = 1
    assert L.yh_debug_read_tensor(eng.h, b"stem", None, 0, C.byref(d4)) == -5
    assert b"debug_tensors" in L.yh_last_error(eng.h)
    assert L.yh_debug_read_tensor(eng.h, b"pool", None, 0, C.byref(d4)) == 0 and d4[3] == 64
    # the error of a failed call does not poison the next good one
    eng.evaluate()
    assert len(eng.detections(0)[0]) >= 0
    eng.close()


def test_classify_argument_checks(built):
    import yolact_amd as ya
    eng = ya.Engine(input_size=100, max_batch=2, use_graph=False)      # 100 % 8 != 0
    eng.load_weights(eng.generate_weights(1))
    frame = np.zeros(64 * 48, np.uint32)
    with pytest.raises(ya.YhError):
        eng.classify_frame(frame, 64, 48, ya.COMPAT_SANE)
    eng.close()
    eng = ya.Engine(input_size=64, max_batch=1, use_graph=False)       # needs two tiles
    eng.load_weights(eng.generate_weights(1))
    with pytest.raises(ya.YhError):
        eng.classify_frame(frame, 64, 48, ya.COMPAT_SANE)
    eng.close()
    eng = ya.Engine(input_size=64, max_batch=2, use_graph=False)
    eng.load_weights(eng.generate_weights(1))
    with pytest.raises(ya.YhError):
        eng.classify_frame(frame, 64, 48, 7)                           # unknown compat mode
    eng.classify_frame(frame, 64, 48, ya.COMPAT_SANE)                  # tiny 64x48 frame works
    assert ((frame >> 24) <= 3).all()
    eng.close()


def test_rccl_weight_broadcast_entry_points_on_one_gpu(built):
    """The library's own RCCL path (yh_rccl_unique_id / yh_rank_broadcast_weights / yh_group_broadcast_weights), as far as
    one GPU lets it run: librccl is opened, a communicator of one rank is created, the broadcast runs and the weights stay
    usable. RCCL refuses two ranks on one device, so n > 1 is the driver's 8-GPU run (bench.py takes this path there, with
    torch.distributed as the fallback if any rank fails)."""
    import numpy as np
    import yolact_amd as ya
    eng = ya.Engine(input_size=128, max_batch=1, use_graph=False, conf_thresh=0.005)
    with pytest.raises(ya.YhError) as e:
        eng.rank_broadcast_weights(b"\0" * 128, 0, 1, 0)          # the root must hold weights first
    assert e.value.code == ya.capi.ESTATE
    blob = eng.generate_weights(seed=1)
    eng.load_weights(blob)
    img = np.random.default_rng(0).integers(0, 256, (1, 128, 128, 3), dtype=np.uint8)
    eng.set_input(img); eng.evaluate()
    before = [eng.output(i) for i in range(4)]
    ident = ya.rccl_unique_id()
    assert len(ident) == 128 and any(ident)
    eng.rank_broadcast_weights(ident, 0, 1, 0)                   # ncclCommInitRank(1 rank) + ncclBroadcast + destroy
    ya.group_broadcast_weights([eng], 0)                          # one handle: nothing to do, still validated
    with pytest.raises(ya.YhError):
        eng.rank_broadcast_weights(ident, 1, 1, 0)               # rank out of range
    assert eng.weights_device_ptr()
    eng.set_input(img); eng.evaluate()
    for a, b in zip(before, [eng.output(i) for i in range(4)]):
        assert np.array_equal(a, b)
    # a second engine on the same device loads straight from the first one's device blob (what a non-root rank does
    # with the bytes it received)
    e2 = ya.Engine(input_size=128, max_batch=1, use_graph=False, conf_thresh=0.005)
    e2.load_weights_device(eng.weights_device_ptr(), eng.weights_nbytes())
    e2.set_input(img); e2.evaluate()
    for a, b in zip(before, [e2.output(i) for i in range(4)]):
        assert np.array_equal(a, b)
    e2.close(); eng.close()
