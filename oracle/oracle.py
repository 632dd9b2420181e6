"""ctypes loader for the CPU oracle (oracle/liboracle.so). TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never
by the product package. See oracle/oracle.h for the parity status of each part.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_ref.c", "orc_detect.c", "orc_net.c", "orc_fp8.c", "orc_scene.c", "oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _proto(_LIB)
    return _LIB


def usable_cores():
    """Host cores this process may really use: the smallest of the machine's count, the affinity mask and the cgroup CPU quota. A
    one-GPU box hands a job 16 of its 256 hardware threads: 256 OpenMP threads inside that quota ran the forward 40 x slower than
    16 (bench.usable_cores, round 2) - and until round 5 that is what every oracle forward of the GPU test suite did."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


class NetCfg(C.Structure):
    _fields_ = [("backbone", C.c_int), ("input_size", C.c_int), ("num_classes", C.c_int)]


class DetCfg(C.Structure):
    _fields_ = [("num_classes", C.c_int), ("top_k", C.c_int), ("max_dets", C.c_int),
                ("conf_thresh", C.c_float), ("nms_thresh", C.c_float)]


class Detection(C.Structure):
    _fields_ = [("class_id", C.c_int32), ("prior", C.c_int32), ("score", C.c_float), ("box", C.c_float * 4)]


def _proto(L):
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orc_unpack_rgb.argtypes = [vp, sz, vp]
    L.orc_pack_rgb.argtypes = [vp, sz, vp]
    L.orc_dequant_u8.argtypes = [vp, sz, f, C.c_int32, vp]
    L.orc_gated_argmax.argtypes = [vp, i, i, vp]
    L.orc_terrible_id.argtypes = [vp, i, i, vp]
    L.orc_terrible_id.restype = i
    L.orc_sane_id.argtypes = [vp, i, i, vp]
    L.orc_postprocess_tile.argtypes = [vp, i, i, i, vp]
    L.orc_postprocess_tile.restype = i
    L.orc_resize_triangle_rgb8.argtypes = [vp, i, i, vp, i, i]
    L.orc_classify_pre.argtypes = [vp, i, i, i, vp]
    L.orc_classify_post.argtypes = [vp, i, i, i, vp, i, i]
    L.orc_classify_post.restype = i
    L.orc_consumer_low16.argtypes = [vp, sz, vp]
    L.orc_weights_nbytes.argtypes = [C.POINTER(NetCfg)]
    L.orc_weights_nbytes.restype = sz
    L.orc_weights_generate.argtypes = [C.POINTER(NetCfg), C.c_uint64, vp, sz]
    L.orc_weights_generate.restype = i
    L.orc_net_create.argtypes = [C.POINTER(NetCfg), vp, sz]
    L.orc_net_create.restype = vp
    L.orc_net_destroy.argtypes = [vp]
    L.orc_net_num_priors.argtypes = [vp]
    L.orc_net_num_priors.restype = i
    L.orc_net_proto_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.orc_net_priors.argtypes = [vp, vp]
    L.orc_net_flops_per_frame.argtypes = [vp]
    L.orc_net_flops_per_frame.restype = C.c_double
    L.orc_net_forward.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp]
    L.orc_net_forward.restype = i
    L.orc_net_get.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(i * 4)]
    L.orc_net_get.restype = C.c_long
    L.orc_conv2d.argtypes = [vp, i, i, i, i, vp, vp, i, i, i, i, i, vp, i, i, i, vp]
    L.orc_bilinear.argtypes = [vp, i, i, i, i, i, i, i, vp]
    L.orc_maxpool3x3s2.argtypes = [vp, i, i, i, i, vp]
    L.orc_f16_round.argtypes = [f]
    L.orc_f16_round.restype = f
    L.orc_f32_to_f16_bits.argtypes = [f]
    L.orc_f32_to_f16_bits.restype = C.c_uint16
    L.orc_f16_bits_to_f32.argtypes = [C.c_uint16]
    L.orc_f16_bits_to_f32.restype = f
    L.orc_spec_expf.argtypes = [f]
    L.orc_spec_expf.restype = f
    L.orc_spec_tanhf.argtypes = [f]
    L.orc_spec_tanhf.restype = f
    L.orc_detect.argtypes = [C.POINTER(DetCfg), vp, vp, vp, vp, vp, i, i, i, vp, vp]
    L.orc_detect.restype = i


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ reference logic
def unpack_rgb(px):
    px = np.ascontiguousarray(px, np.uint32)
    out = np.empty(px.size * 3, np.uint8)
    lib().orc_unpack_rgb(_p(px), px.size, _p(out))
    return out


def pack_rgb(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1)
    out = np.empty(rgb.size // 3, np.uint32)
    lib().orc_pack_rgb(_p(rgb), rgb.size // 3, _p(out))
    return out


def dequant_u8(q, scale, zero_point):
    q = np.ascontiguousarray(q, np.uint8)
    out = np.empty(q.size, np.float32)
    lib().orc_dequant_u8(_p(q), q.size, scale, zero_point, _p(out))
    return out


def gated_argmax(dets, nch=81):
    dets = np.ascontiguousarray(dets, np.float32).reshape(-1)
    n = dets.size // nch
    out = np.empty(n, np.uint8)
    lib().orc_gated_argmax(_p(dets), n, nch, _p(out))
    return out


def terrible_id(classes, grid_w=28):
    classes = np.ascontiguousarray(classes, np.uint8).reshape(-1)
    out = np.empty(classes.size, np.int8)
    rc = lib().orc_terrible_id(_p(classes), classes.size, grid_w, _p(out))
    return rc, out


def sane_id(classes, gh, gw):
    classes = np.ascontiguousarray(classes, np.uint8).reshape(-1)
    out = np.empty(classes.size, np.int8)
    lib().orc_sane_id(_p(classes), gh, gw, _p(out))
    return out


def postprocess_tile(cells, grid=28, nch=81, mode=0):
    cells = np.ascontiguousarray(cells, np.float32).reshape(-1)
    out = np.zeros((grid * 8) ** 2, np.uint32)
    rc = lib().orc_postprocess_tile(_p(cells), grid, nch, mode, _p(out))
    return rc, out


def resize_triangle_rgb8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape[:2]
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_triangle_rgb8(_p(src), sw, sh, _p(out), dw, dh)
    return out


def classify_pre(frame, w, h, S):
    frame = np.ascontiguousarray(frame, np.uint32).reshape(-1)
    out = np.empty((2, S, S, 3), np.uint8)
    lib().orc_classify_pre(_p(frame), w, h, S, _p(out))
    return out


def classify_post(cells2, S, nch, mode, w, h):
    cells2 = np.ascontiguousarray(cells2, np.float32).reshape(-1)
    frame = np.zeros(w * h, np.uint32)
    rc = lib().orc_classify_post(_p(cells2), S, nch, mode, _p(frame), w, h)
    return rc, frame


def consumer_low16(frame):
    frame = np.ascontiguousarray(frame, np.uint32).reshape(-1)
    out = np.empty(frame.size, np.uint16)
    lib().orc_consumer_low16(_p(frame), frame.size, _p(out))
    return out


# ------------------------------------------------------------------ network
class Net:
    def __init__(self, backbone=50, input_size=550, num_classes=81, seed=1, blob=None):
        self.cfg = NetCfg(backbone, input_size, num_classes)
        L = lib()
        nb = L.orc_weights_nbytes(C.byref(self.cfg))
        if blob is None:
            blob = np.zeros(nb, np.uint8)
            assert L.orc_weights_generate(C.byref(self.cfg), seed, _p(blob), nb) == 0
        self.blob = np.ascontiguousarray(blob, np.uint8)
        assert self.blob.size == nb, (self.blob.size, nb)
        self.h = L.orc_net_create(C.byref(self.cfg), _p(self.blob), nb)
        assert self.h, "orc_net_create failed"
        self.P = L.orc_net_num_priors(self.h)
        hp, wp = C.c_int(), C.c_int()
        L.orc_net_proto_dims(self.h, C.byref(hp), C.byref(wp))
        self.hp, self.wp = hp.value, wp.value
        self.C = num_classes
        self.S = input_size

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_net_destroy(self.h)
            self.h = None

    def priors(self):
        out = np.empty((self.P, 4), np.float32)
        lib().orc_net_priors(self.h, _p(out))
        return out

    def flops_per_frame(self):
        return lib().orc_net_flops_per_frame(self.h)

    def set_fp8(self, layers=None):
        """fp8 forward mode (configs[4]): `layers` = {conv name: activation scales} of the convolutions that read E4M3
        operands (names as in orc_net.c: l3b0_b, p5, proto0, head_t0 ...) - one scale per INPUT CHANNEL (an array), or a
        single number for the same scale in every channel; None / {} switches it off."""
        L = lib()
        L.orc_net_clear_fp8.argtypes = [C.c_void_p]
        L.orc_net_add_fp8_layer_ch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int]
        L.orc_net_add_fp8_layer_ch.restype = C.c_int
        L.orc_net_clear_fp8(self.h)
        for name, scale in (layers or {}).items():
            v = np.ascontiguousarray(np.atleast_1d(np.asarray(scale, np.float32)))
            assert L.orc_net_add_fp8_layer_ch(self.h, name.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), int(v.size)) == 0, name

    def set_fp8_study_ex(self, act_mode=0, w_mode=1, skip=""):
        """Extended accuracy study: act_mode 0 off, 1 per tensor, 2 E8M0 blocks of 32, 3 per input channel;
        w_mode 1 per output channel, 2 E8M0 blocks of 32; skip: comma-separated conv-name prefixes kept in f16."""
        L = lib()
        L.orc_net_set_fp8_study_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.orc_net_set_fp8_study_ex.restype = None
        L.orc_net_set_fp8_study_ex(self.h, int(act_mode), int(w_mode), skip.encode())

    def forward(self, rgb, f16=True, nthreads=None, fp8_study=False):
        """fp8_study: the K-heavy 3x3 convs run on E4M3-rounded operands (accuracy study)."""
        lib().orc_net_set_fp8_study.argtypes = [C.c_void_p, C.c_int]
        lib().orc_net_set_fp8_study.restype = None
        lib().orc_net_set_fp8_study(self.h, int(fp8_study))   # 0 off, 1 per-tensor / per-channel scales, 2 MX blocks of 32
        rgb = np.ascontiguousarray(rgb, np.uint8)
        n = rgb.shape[0]
        assert rgb.shape == (n, self.S, self.S, 3)
        loc = np.empty((n, self.P, 4), np.float32)
        conf = np.empty((n, self.P, self.C), np.float32)
        mask = np.empty((n, self.P, 32), np.float32)
        proto = np.empty((n, self.hp, self.wp, 32), np.float32)
        nt = nthreads or usable_cores()
        rc = lib().orc_net_forward(self.h, _p(rgb), n, 1 if f16 else 0, nt, _p(loc), _p(conf), _p(mask), _p(proto))
        assert rc == 0
        return loc, conf, mask, proto

    def get(self, name):
        dims = (C.c_int * 4)()
        ne = lib().orc_net_get(self.h, name.encode(), None, 0, C.byref(dims))
        if ne < 0:
            raise KeyError(name)
        out = np.empty(tuple(dims), np.float32)
        lib().orc_net_get(self.h, name.encode(), _p(out), out.size, C.byref(dims))
        return out


def conv2d(x, w, bias, stride=1, pad=0, residual=None, act=0, f16=False, nthreads=None):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    n, h, ww, cin = x.shape
    cout, kh, kw, _ = w.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (ww + 2 * pad - kw) // stride + 1
    y = np.empty((n, ho, wo, cout), np.float32)
    res = None if residual is None else np.ascontiguousarray(residual, np.float32)
    lib().orc_conv2d(_p(x), n, h, ww, cin, _p(w), _p(bias), cout, kh, kw, stride, pad,
                     _p(res) if res is not None else None, act, 1 if f16 else 0,
                     nthreads or usable_cores(), _p(y))
    return y


def bilinear(x, ho, wo, f16=False):
    x = np.ascontiguousarray(x, np.float32)
    n, h, w, c = x.shape
    y = np.empty((n, ho, wo, c), np.float32)
    lib().orc_bilinear(_p(x), n, h, w, c, ho, wo, 1 if f16 else 0, _p(y))
    return y


def maxpool3x3s2(x):
    x = np.ascontiguousarray(x, np.float32)
    n, h, w, c = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = np.empty((n, ho, wo, c), np.float32)
    lib().orc_maxpool3x3s2(_p(x), n, h, w, c, _p(y))
    return y


def f16_round(a):
    a = np.asarray(a, np.float32)
    L = lib()
    return np.array([L.orc_f16_round(float(v)) for v in a.reshape(-1)], np.float32).reshape(a.shape)


def spec_expf(x):
    return lib().orc_spec_expf(float(x))


def spec_tanhf(x):
    return lib().orc_spec_tanhf(float(x))


def detect(loc, conf, mask, proto, priors, num_classes=81, top_k=200, max_dets=100,
           conf_thresh=0.05, nms_thresh=0.5, want_masks=True):
    """One frame. Returns (list of dict, masks[nd,hp,wp] uint8)."""
    loc = np.ascontiguousarray(loc, np.float32)
    conf = np.ascontiguousarray(conf, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    proto = np.ascontiguousarray(proto, np.float32)
    priors = np.ascontiguousarray(priors, np.float32)
    P = priors.shape[0]
    hp, wp = proto.shape[0], proto.shape[1]
    cfg = DetCfg(num_classes, top_k, max_dets, conf_thresh, nms_thresh)
    dets = (Detection * max_dets)()
    masks = np.zeros((max_dets, hp, wp), np.uint8) if want_masks else None
    nd = lib().orc_detect(C.byref(cfg), _p(loc), _p(conf), _p(mask), _p(proto), _p(priors), P, hp, wp,
                          C.cast(dets, C.c_void_p), _p(masks) if want_masks else None)
    out = [dict(class_id=d.class_id, prior=d.prior, score=d.score, box=tuple(d.box)) for d in dets[:nd]]
    return out, (masks[:nd] if want_masks else None)


# ---- OCP FP8 E4M3 (orc_fp8.c) --------------------------------------------------------------------
def e4m3_decode_table():
    """The 256 decoded values (NaN at 0x7F / 0xFF)."""
    L = lib()
    L.orc_e4m3_to_f32.restype = C.c_float
    L.orc_e4m3_to_f32.argtypes = [C.c_uint8]
    return np.array([L.orc_e4m3_to_f32(b) for b in range(256)], np.float32)


def quantize_e4m3(x, inv_scale=1.0):
    """uint8 codes of round-to-nearest-even, saturating e4m3 of x * inv_scale (f32 multiply)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.shape, np.uint8)
    L = lib()
    L.orc_quantize_e4m3.argtypes = [C.c_void_p, C.c_longlong, C.c_float, C.c_void_p]
    L.orc_quantize_e4m3.restype = None
    L.orc_quantize_e4m3(_p(x), x.size, C.c_float(inv_scale), _p(y))
    return y


# ---- scene back-end (orc_scene.c): shaders/pt_cloud.comp + pt_cloud_weights.comp ---------------------------
def spec_logf(x):
    L = lib()
    L.orc_spec_logf.restype = C.c_float
    L.orc_spec_logf.argtypes = [C.c_float]
    return L.orc_spec_logf(float(x))


def scene(depth, cls_id, mode=0):
    """depth [H][W] uint16, cls_id [H][W][2] uint8 (class, id) -> dict(map u32 [H][W], world / conn0 / conn1 f32 [H][W][4],
    balls f32 [100][4]). mode 0 STRICT (pack() uses `&`), 1 SANE."""
    depth = np.ascontiguousarray(depth, np.uint16)
    cls_id = np.ascontiguousarray(cls_id, np.uint8)
    H, W = depth.shape
    assert cls_id.shape == (H, W, 2)
    out = dict(map=np.zeros((H, W), np.uint32), world=np.zeros((H, W, 4), np.float32), conn0=np.zeros((H, W, 4), np.float32),
               conn1=np.zeros((H, W, 4), np.float32), balls=np.zeros((100, 4), np.float32))
    L = lib()
    L.orc_scene.restype = None
    L.orc_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.orc_scene(_p(depth), _p(cls_id), W, H, mode, _p(out["map"]), _p(out["world"]), _p(out["conn0"]), _p(out["conn1"]), _p(out["balls"]))
    return out
