"""ctypes binding of include/yolact_hip.h (the C ABI of libyolact_hip.so)."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

OK, EINVAL, EHIP, ENOMEM, EWEIGHTS, ESTATE, EDIVERGE, EOVERFLOW = 0, -1, -2, -3, -4, -5, -6, -7
COMPAT_STRICT, COMPAT_SANE = 0, 1


class YhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"yolact_hip error {code}: {msg}")
        self.code = code


PRECISION_F16, PRECISION_FP8 = 0, 1

# yh_tuning (include/yolact_hip.h): per-handle measurement / test knobs, -1 = the library's default
TUNING_FIELDS = ("plan_cus", "mfma16", "t128x256_m16", "small16", "bigk", "tailsplit", "chsplit", "k1tile", "k1_maxk",
                 "splitk_minsteps", "t64", "t64_maxb", "t64_minsteps", "stemfuse", "prefuse", "headmerge",
                 "upfuse", "k1_generic", "ablate", "op_tile", "op_kslices", "tfl_dot", "tfl_graph", "tailfork", "dsfuse", "headfork_maxb", "protofuse", "k1_min1", "k1_min3", "chain", "tfl_fuse", "tfl_group")


class Tuning(C.Structure):
    _fields_ = [(f, C.c_int32) for f in TUNING_FIELDS]

    @classmethod
    def of(cls, **kw):
        t = cls()
        C.memset(C.byref(t), 0xFF, C.sizeof(t))
        for k, v in kw.items():
            assert k in TUNING_FIELDS, k
            setattr(t, k, int(v))
        return t

    def as_dict(self):
        return {f: getattr(self, f) for f in TUNING_FIELDS}


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("backbone", C.c_int32),
                ("input_size", C.c_int32), ("max_batch", C.c_int32), ("num_classes", C.c_int32),
                ("top_k", C.c_int32), ("max_dets", C.c_int32), ("conf_thresh", C.c_float),
                ("nms_thresh", C.c_float), ("use_graph", C.c_int32), ("debug_tensors", C.c_int32),
                ("precision", C.c_int32), ("fp8_f16_layers", C.c_int32), ("fp8_per_tensor", C.c_int32), ("reserved", C.c_int32 * 4), ("tune", Tuning)]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("ndims", C.c_int32), ("dims", C.c_int32 * 4),
                ("scale", C.c_float), ("zero_point", C.c_int32)]


class Detection(C.Structure):
    _fields_ = [("class_id", C.c_int32), ("prior", C.c_int32), ("score", C.c_float), ("box", C.c_float * 4)]


def lib_path():
    return os.path.join(_PKG, "lib", "libyolact_hip.so")


# every symbol include/yolact_hip.h declares: (name, restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
SYMBOLS = [
    ("yh_version", C.c_char_p, []),
    ("yh_default_config", None, [C.POINTER(Config)]),
    ("yh_create", _i, [C.POINTER(Config), C.POINTER(_vp)]),
    ("yh_destroy", None, [_vp]),
    ("yh_set_tuning", _i, [_vp, C.POINTER(Tuning)]),
    ("yh_get_tuning", _i, [_vp, C.POINTER(Tuning)]),
    ("yh_last_error", C.c_char_p, [_vp]),
    ("yh_weights_nbytes", _sz, [_vp]),
    ("yh_weights_device_ptr", _vp, [_vp]),
    ("yh_weights_generate", _i, [_vp, C.c_uint64, _vp, _sz]),
    ("yh_load_weights_host", _i, [_vp, _vp, _sz]),
    ("yh_load_weights_device", _i, [_vp, _vp, _sz]),
    ("yh_fp8_calibrate", _i, [_vp]),
    ("yh_fp8_layer_count", _i, [_vp]),
    ("yh_fp8_layer_info", _i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_f)]),
    ("yh_fp8_set_layer_scale", _i, [_vp, _i, _f]),
    ("yh_fp8_layer_channels", _i, [_vp, _i]),
    ("yh_fp8_layer_channel_scales", _i, [_vp, _i, _vp, _i]),
    ("yh_fp8_set_layer_channel_scales", _i, [_vp, _i, _vp, _i]),
    ("yh_group_broadcast_weights", _i, [C.POINTER(_vp), _i, _i]),
    ("yh_rccl_unique_id", _i, [_vp]),
    ("yh_rank_broadcast_weights", _i, [_vp, _vp, _i, _i, _i]),
    ("yh_group_create", _i, [C.POINTER(Config), C.POINTER(_i), _i, C.POINTER(_vp)]),
    ("yh_group_destroy", None, [_vp]),
    ("yh_group_last_error", C.c_char_p, [_vp]),
    ("yh_group_size", _i, [_vp]),
    ("yh_group_member", _vp, [_vp, _i]),
    ("yh_group_load_weights_host", _i, [_vp, _vp, _sz]),
    ("yh_group_replicate_weights", _i, [_vp]),
    ("yh_group_weights_replication", C.c_char_p, [_vp]),
    ("yh_group_fp8_calibrate", _i, [_vp]),
    ("yh_group_evaluate", _i, [_vp, _vp, _i, _i]),
    ("yh_group_evaluate_device", _i, [_vp, C.POINTER(_vp), C.POINTER(_i), _i]),
    ("yh_group_prepare", _i, [_vp, _i, _i]),
    ("yh_group_sync", _i, [_vp]),
    ("yh_group_read_detections", _i, [_vp, _i, C.POINTER(_i), _vp, _i, _vp, _sz]),
    ("yh_group_frame_owner", _i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("yh_input_dims", _i, [_vp, C.POINTER(_i * 4)]),
    ("yh_set_input_u8", _i, [_vp, _vp, _i]),
    ("yh_set_input_u8_device", _i, [_vp, _vp, _i]),
    ("yh_invoke", _i, [_vp]),
    ("yh_prepare", _i, [_vp, _i, _i]),
    ("yh_sync", _i, [_vp]),
    ("yh_output_count", _i, [_vp]),
    ("yh_output_info", _i, [_vp, _i, C.POINTER(TensorInfo)]),
    ("yh_output_read_f32", _i, [_vp, _i, _vp, _sz]),
    ("yh_output_device_ptr", _vp, [_vp, _i]),
    ("yh_evaluate", _i, [_vp]),
    ("yh_read_detections", _i, [_vp, _i, C.POINTER(_i), _vp, _i, _vp, _sz]),
    ("yh_proto_dims", _i, [_vp, C.POINTER(_i * 2)]),
    ("yh_num_priors", _i, [_vp]),
    ("yh_read_priors", _i, [_vp, _vp, _sz]),
    ("yh_classify_frame_u32", _i, [_vp, _vp, _i, _i, _i]),
    ("yh_postprocess_cells", _i, [_vp, _vp, _i, _vp, _i]),
    ("yh_resize_triangle_rgb8", _i, [_vp, _vp, _i, _i, _vp, _i, _i]),
    ("yh_profile_launch_count", _i, [_vp, _i]),
    ("yh_profile_run", _i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("yh_time_steps", _i, [_vp, _i, _i, C.POINTER(_f)]),
    ("yh_flops_per_frame", C.c_double, [_vp]),
    ("yh_tfl_validate", _i, [_vp, _sz, C.POINTER(_i), C.POINTER(_i), C.c_char_p, _sz]),
    ("yh_tfl_create", _i, [_vp, _sz, _i, C.POINTER(_vp)]),
    ("yh_tfl_create_tuned", _i, [_vp, _sz, _i, C.POINTER(Tuning), C.POINTER(_vp)]),
    ("yh_tfl_destroy", None, [_vp]),
    ("yh_tfl_last_error", C.c_char_p, [_vp]),
    ("yh_tfl_input_info", _i, [_vp, C.POINTER(TensorInfo)]),
    ("yh_tfl_output_count", _i, [_vp]),
    ("yh_tfl_output_info", _i, [_vp, _i, C.POINTER(TensorInfo)]),
    ("yh_tfl_set_batch", _i, [_vp, _i]),
    ("yh_tfl_set_input", _i, [_vp, _vp, _sz]),
    ("yh_tfl_invoke", _i, [_vp]),
    ("yh_tfl_output_read", _i, [_vp, _i, _vp, _sz]),
    ("yh_tfl_tensor_count", _i, [_vp]),
    ("yh_tfl_plan_info", _i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("yh_tfl_tensor_read", _i, [_vp, _i, _vp, _sz]),
    ("yh_tfl_classify_frame_u32", _i, [_vp, _vp, _i, _i, _i]),
    ("yh_scene_create", _i, [_i, _i, _i, C.POINTER(_vp)]),
    ("yh_scene_destroy", None, [_vp]),
    ("yh_scene_last_error", C.c_char_p, [_vp]),
    ("yh_scene_append", _i, [_vp, _vp, _vp, _i]),
    ("yh_scene_append_classified", _i, [_vp, _vp, _vp, _i, _i]),
    ("yh_scene_read", _i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("yh_scene_time", _i, [_vp, _i, C.POINTER(_f)]),
    ("yh_classify_device_frame", _vp, [_vp]),
    ("yh_debug_read_tensor", _i, [_vp, C.c_char_p, _vp, _sz, C.POINTER(_i * 4)]),
    ("yh_debug_read_tensor_frame", _i, [_vp, C.c_char_p, _i, _vp, _sz, C.POINTER(_i * 4)]),
    ("yh_debug_last_conv_launches", _i, [_vp]),
    ("yh_debug_alloc_map", _i, [_vp, C.c_char_p, _sz]),
    ("yh_debug_setup_audit", _i, [C.POINTER(C.c_int64 * 4)]),
    ("yh_debug_rccl_shared_device", _i, [_i]),
    ("yh_debug_rccl_library", _i, [C.c_char_p]),
    ("yh_debug_set_cu_mask", _i, [_vp, _vp, _i]),
    ("yh_debug_run_phase", _i, [_vp, _i, _i, C.POINTER(_f)]),
    ("yh_debug_graph_nodes", _i, [_vp, _i, C.c_char_p, _sz]),
    ("yh_op_stem_pool_f16", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("yh_op_stem_pool_rgb8", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("yh_op_quantize_e4m3", _i, [_vp, _vp, _sz, C.c_float, _vp]),
    ("yh_op_conv2d_fp8", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    ("yh_op_conv2d_dual_f16", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    ("yh_op_conv2d_levels_f16", _i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    ("yh_op_conv2d_f16", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    ("yh_op_bilinear_f16", _i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    ("yh_op_maxpool3x3s2_f16", _i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    ("yh_op_detect", _i, [_vp, _vp, _vp, _vp, _vp, _i]),
]


def load_library():
    """Loads libyolact_hip.so. Raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(there is no CPU fallback for the HIP kernel library)")
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the ABI symbol is absent
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def version():
    return load_library().yh_version().decode()


def setup_audit():
    """yh_debug_setup_audit: process-wide counters of the setup discipline -
    dict(setups, worker_jobs, overlaps, in_flight); `overlaps` must stay 0."""
    L = load_library()
    a = (C.c_int64 * 4)()
    rc = L.yh_debug_setup_audit(C.byref(a))
    if rc != OK:
        raise YhError(rc, "yh_debug_setup_audit")
    return dict(setups=a[0], worker_jobs=a[1], overlaps=a[2], in_flight=a[3])


def rccl_unique_id():
    """ncclGetUniqueId through the library (128 bytes) for yh_rank_broadcast_weights."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.yh_rccl_unique_id(buf)
    if rc != OK:
        raise YhError(rc, L.yh_last_error(None).decode())
    return buf.raw


def group_broadcast_weights(engines, root=0):
    """One process, one Engine per GPU: RCCL broadcast of engines[root]'s weights to the others."""
    L = load_library()
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    rc = L.yh_group_broadcast_weights(arr, len(engines), root)
    if rc != OK:
        raise YhError(rc, L.yh_last_error(engines[root].h).decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f16_bits(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float16)).view(np.uint16)


def _bits_f32(b, shape):
    return b.view(np.float16).astype(np.float32).reshape(shape)


class Engine:
    """RAII wrapper of one yh_engine handle."""

    def __init__(self, input_size=550, backbone=50, max_batch=1, num_classes=81, top_k=200, max_dets=100,
                 conf_thresh=0.05, nms_thresh=0.5, use_graph=True, device=0, debug_tensors=False, precision=PRECISION_F16,
                 tune=None, fp8_f16_layers=0, fp8_per_tensor=False):
        self.L = load_library()
        cfg = Config()
        self.L.yh_default_config(C.byref(cfg))
        cfg.device, cfg.backbone, cfg.input_size, cfg.max_batch = device, backbone, input_size, max_batch
        cfg.num_classes, cfg.top_k, cfg.max_dets = num_classes, top_k, max_dets
        cfg.conf_thresh, cfg.nms_thresh, cfg.use_graph = conf_thresh, nms_thresh, 1 if use_graph else 0
        cfg.debug_tensors = 1 if debug_tensors else 0
        cfg.precision = precision
        cfg.fp8_f16_layers = fp8_f16_layers
        cfg.fp8_per_tensor = 1 if fp8_per_tensor else 0
        if tune:
            cfg.tune = Tuning.of(**tune)
        self._tune = dict(tune or {})
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.L.yh_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise YhError(rc, self.L.yh_last_error(None).decode())
        self.h = h
        self.S, self.C, self.max_batch = input_size, num_classes, max_batch
        self.P = self.L.yh_num_priors(self.h)
        d = (C.c_int32 * 2)()
        self.L.yh_proto_dims(self.h, C.byref(d))
        self.hp, self.wp = d[0], d[1]
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.yh_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise YhError(rc, self.L.yh_last_error(self.h).decode())

    # ---- tuning (per handle; the library reads no environment variable)
    def set_tuning(self, **kw):
        """Replaces the run-time tuning fields; fields not named keep what this wrapper last set."""
        self._tune.update(kw)
        t = Tuning.of(**self._tune)
        self._chk(self.L.yh_set_tuning(self.h, C.byref(t)))

    def reset_tuning(self, *names):
        for k in names:
            self._tune.pop(k, None)
        t = Tuning.of(**self._tune)
        self._chk(self.L.yh_set_tuning(self.h, C.byref(t)))

    def tuning(self):
        t = Tuning()
        self._chk(self.L.yh_get_tuning(self.h, C.byref(t)))
        return t.as_dict()

    # ---- weights
    def weights_nbytes(self):
        return self.L.yh_weights_nbytes(self.h)

    def generate_weights(self, seed=1):
        blob = np.zeros(self.weights_nbytes(), np.uint8)
        self._chk(self.L.yh_weights_generate(self.h, seed, _p(blob), blob.size))
        return blob

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        self._chk(self.L.yh_load_weights_host(self.h, _p(blob), blob.size))

    def weights_device_ptr(self):
        return self.L.yh_weights_device_ptr(self.h)

    def load_weights_device(self, dev_ptr, nbytes):
        self._chk(self.L.yh_load_weights_device(self.h, C.c_void_p(dev_ptr), nbytes))

    # ---- fp8 precision (configs[4])
    def fp8_calibrate(self):
        """Sets every fp8 input tensor's scale from the f16 forward of the frames last set."""
        self._chk(self.L.yh_fp8_calibrate(self.h))

    def fp8_layers(self):
        """[(conv name, activation scale)] of the convolutions that read E4M3 operands, in execution order."""
        out = []
        for i in range(self.L.yh_fp8_layer_count(self.h)):
            name, sc = C.c_char_p(), C.c_float()
            self._chk(self.L.yh_fp8_layer_info(self.h, i, C.byref(name), C.byref(sc)))
            out.append((name.value.decode(), sc.value))
        return out

    def fp8_set_layer_scale(self, i, scale):
        """One number: the same scale in every channel of layer i's input tensor; an array: one per channel."""
        if np.ndim(scale) == 0:
            self._chk(self.L.yh_fp8_set_layer_scale(self.h, i, C.c_float(scale)))
        else:
            v = np.ascontiguousarray(scale, np.float32)
            self._chk(self.L.yh_fp8_set_layer_channel_scales(self.h, i, _p(v), int(v.size)))

    def fp8_channel_scales(self):
        """[(conv name, per-input-channel activation scales as an f32 array)] of the E4M3 convolutions, in execution order -
        what the oracle's fp8 forward mode is given (oracle.Net.set_fp8) and what a host stores with the model."""
        out = []
        for i, (name, _) in enumerate(self.fp8_layers()):
            nc = self.L.yh_fp8_layer_channels(self.h, i)
            if nc < 1:
                raise YhError(nc, "yh_fp8_layer_channels")
            v = np.empty(nc, np.float32)
            self._chk(self.L.yh_fp8_layer_channel_scales(self.h, i, _p(v), nc))
            out.append((name, v))
        return out

    def rank_broadcast_weights(self, id_bytes, rank, nranks, root=0):
        """One process per GPU: RCCL broadcast of the root rank's weights (id_bytes from rccl_unique_id on one rank)."""
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        self._chk(self.L.yh_rank_broadcast_weights(self.h, buf, rank, nranks, root))

    # ---- interpreter-shaped surface
    def input_dims(self):
        d = (C.c_int32 * 4)()
        self._chk(self.L.yh_input_dims(self.h, C.byref(d)))
        return tuple(d)

    def set_input(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        n = rgb.shape[0]
        assert rgb.shape == (n, self.S, self.S, 3), rgb.shape
        self._chk(self.L.yh_set_input_u8(self.h, _p(rgb), n))
        self.n = n

    def set_input_device(self, dev_ptr, n):
        self._chk(self.L.yh_set_input_u8_device(self.h, C.c_void_p(dev_ptr), n))
        self.n = n

    def invoke(self):
        self._chk(self.L.yh_invoke(self.h))

    def evaluate(self):
        self._chk(self.L.yh_evaluate(self.h))

    def set_cu_mask(self, n_cus, total=256, offset=0):
        """Study hook: the handle's compute streams on CUs offset .. offset + n_cus - 1 of `total` (yh_debug_set_cu_mask)."""
        words = (total + 31) // 32
        m = (C.c_uint32 * words)()
        for i in range(offset, offset + n_cus):
            m[i // 32] |= 1 << (i % 32)
        self._chk(self.L.yh_debug_set_cu_mask(self.h, C.cast(m, C.c_void_p), words))

    def run_phase(self, phase, reps=1):
        """Study hook: device ms of `reps` runs of one phase of the forward (yh_debug_run_phase)."""
        ms = C.c_float()
        self._chk(self.L.yh_debug_run_phase(self.h, phase, reps, C.byref(ms)))
        return ms.value

    def prepare(self, n, with_tail=True):
        """yh_prepare: capture the step for n frames now, on this thread (allocate_tensors() at its proper time)."""
        self._chk(self.L.yh_prepare(self.h, n, 1 if with_tail else 0))

    def sync(self):
        self._chk(self.L.yh_sync(self.h))

    def output_count(self):
        return self.L.yh_output_count(self.h)

    def output_info(self, i):
        ti = TensorInfo()
        self._chk(self.L.yh_output_info(self.h, i, C.byref(ti)))
        return dict(name=ti.name.decode(), kind=ti.kind, dims=tuple(ti.dims[:ti.ndims]), scale=ti.scale,
                    zero_point=ti.zero_point)

    def output(self, i):
        info = self.output_info(i)
        out = np.empty(info["dims"], np.float32)
        self._chk(self.L.yh_output_read_f32(self.h, i, _p(out), out.size))
        return out

    def priors(self):
        out = np.empty((self.P, 4), np.float32)
        self._chk(self.L.yh_read_priors(self.h, _p(out), out.size))
        return out

    def detections(self, frame, want_masks=True):
        nd = C.c_int32()
        dets = (Detection * self.cfg.max_dets)()
        masks = np.zeros((self.cfg.max_dets, self.hp, self.wp), np.uint8) if want_masks else None
        self._chk(self.L.yh_read_detections(self.h, frame, C.byref(nd), C.cast(dets, C.c_void_p), self.cfg.max_dets,
                                            _p(masks) if want_masks else None, masks.size if want_masks else 0))
        out = [dict(class_id=d.class_id, prior=d.prior, score=d.score, box=tuple(d.box)) for d in dets[:nd.value]]
        return out, (masks[:nd.value] if want_masks else None)

    def flops_per_frame(self):
        return self.L.yh_flops_per_frame(self.h)

    # ---- reference-compat path
    def classify_device_frame(self):
        """Device pointer of the frame the last classify_frame produced (for Scene.append_classified)."""
        return self.L.yh_classify_device_frame(self.h)

    def classify_frame(self, frame_u32, width, height, mode=COMPAT_STRICT):
        """In place on a C-contiguous uint32 array of width*height packed pixels."""
        assert frame_u32.dtype == np.uint32 and frame_u32.flags.c_contiguous and frame_u32.size == width * height
        self._chk(self.L.yh_classify_frame_u32(self.h, _p(frame_u32), width, height, mode))

    def postprocess_cells(self, cells, mode=COMPAT_STRICT):
        cells = np.ascontiguousarray(cells, np.float32)
        n = cells.shape[0]
        out = np.zeros((n, self.S, self.S), np.uint32)
        self._chk(self.L.yh_postprocess_cells(self.h, _p(cells), n, _p(out), mode))
        return out

    def resize_triangle(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        sh, sw = src.shape[:2]
        out = np.empty((dh, dw, 3), np.uint8)
        self._chk(self.L.yh_resize_triangle_rgb8(self.h, _p(src), sw, sh, _p(out), dw, dh))
        return out

    def tensor(self, name):
        """Named intermediate of the last forward as f32 NHWC (test hook)."""
        d = (C.c_int32 * 4)()
        self._chk(self.L.yh_debug_read_tensor(self.h, name.encode(), None, 0, C.byref(d)))
        out = np.empty(tuple(d), np.float32)
        self._chk(self.L.yh_debug_read_tensor(self.h, name.encode(), _p(out), out.size, C.byref(d)))
        return out

    def tensor_frame(self, name, frame):
        """One frame of a named intermediate as f32 [h][w][c] (test hook)."""
        d = (C.c_int32 * 4)()
        self._chk(self.L.yh_debug_read_tensor_frame(self.h, name.encode(), frame, None, 0, C.byref(d)))
        out = np.empty(tuple(d)[1:], np.float32)
        self._chk(self.L.yh_debug_read_tensor_frame(self.h, name.encode(), frame, _p(out), out.size, C.byref(d)))
        return out

    # ---- measurement hooks
    def profile(self, with_tail=True, reps=5):
        nl = self.L.yh_profile_launch_count(self.h, 1 if with_tail else 0)
        ms = np.zeros(nl, np.float32)
        fl = np.zeros(nl, np.float64)
        by = np.zeros(nl, np.float64)
        names = (C.c_char_p * nl)()
        self._chk(self.L.yh_profile_run(self.h, 1 if with_tail else 0, reps, _p(ms), _p(fl), _p(by), C.cast(names, C.c_void_p)))
        return [dict(name=names[i].decode(), ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i])) for i in range(nl)]

    def time_steps(self, steps, with_tail=True):
        ms = C.c_float()
        self._chk(self.L.yh_time_steps(self.h, 1 if with_tail else 0, steps, C.byref(ms)))
        return ms.value

    def alloc_map(self):
        """Audit hook: one text line per buffer of the handle (base, end, size, offsets inside their 2 MiB pages)."""
        buf = C.create_string_buffer(1 << 20)
        self._chk(self.L.yh_debug_alloc_map(self.h, buf, len(buf)))
        return buf.value.decode()

    def graph_nodes(self, with_tail=True):
        """Audit hook: the captured step for the current batch size, one sorted text line per graph node."""
        buf = C.create_string_buffer(1 << 20)
        self._chk(self.L.yh_debug_graph_nodes(self.h, 1 if with_tail else 0, buf, len(buf)))
        return buf.value.decode()

    def last_conv_launches(self):
        return self.L.yh_debug_last_conv_launches(self.h)

    # ---- single ops (tests)
    def op_conv2d(self, x, w, bias, stride=1, pad=0, residual=None, act=0):
        n, hh, ww, cin = x.shape
        cout, kh, kw, _ = w.shape
        ho, wo = (hh + 2 * pad - kh) // stride + 1, (ww + 2 * pad - kw) // stride + 1
        xb, wb = _f16_bits(x), _f16_bits(w)
        bias = np.ascontiguousarray(bias, np.float32)
        rb = _f16_bits(residual) if residual is not None else None
        y = np.zeros((n, ho, wo, cout), np.uint16)
        self._chk(self.L.yh_op_conv2d_f16(self.h, _p(xb), n, hh, ww, cin, _p(wb), _p(bias), cout, kh, kw, stride, pad,
                                          _p(rb) if rb is not None else None, act, _p(y)))
        return _bits_f32(y, y.shape)

    def op_bilinear(self, x, ho, wo):
        n, hh, ww, c = x.shape
        xb = _f16_bits(x)
        y = np.zeros((n, ho, wo, c), np.uint16)
        self._chk(self.L.yh_op_bilinear_f16(self.h, _p(xb), n, hh, ww, c, ho, wo, _p(y)))
        return _bits_f32(y, y.shape)

    def op_maxpool(self, x):
        n, hh, ww, c = x.shape
        ho, wo = (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1
        xb = _f16_bits(x)
        y = np.zeros((n, ho, wo, c), np.uint16)
        self._chk(self.L.yh_op_maxpool3x3s2_f16(self.h, _p(xb), n, hh, ww, c, _p(y)))
        return _bits_f32(y, y.shape)

    def op_conv2d_dual(self, x1, x2, stride2, w, bias, act=0):
        """The two-source 1x1 conv: x1 [n][ho][wo][c1], x2 [n][h2][w2][c2] read at stride2, w [cout][c1 + c2] -> [n][ho][wo][cout] f32."""
        n, ho, wo, c1 = x1.shape
        _, h2, w2, c2 = x2.shape
        cout = w.shape[0]
        assert w.shape[1] == c1 + c2
        x1b, x2b, wb = _f16_bits(x1), _f16_bits(x2), _f16_bits(w)
        bias = np.ascontiguousarray(bias, np.float32)
        y = np.zeros((n, ho, wo, cout), np.uint16)
        self._chk(self.L.yh_op_conv2d_dual_f16(self.h, _p(x1b), n, ho, wo, c1, _p(x2b), h2, w2, c2, stride2, _p(wb), _p(bias), cout, act, _p(y)))
        return y.view(np.float16).astype(np.float32)

    def op_conv2d_levels(self, x, level_sizes, w, bias, act=0):
        """x [n][cells][cin] with cells = sum(s*s for s in level_sizes); w [cout][k][k][cin] -> [n][cells][cout] f32."""
        n, cells, cin = x.shape
        cout, k = w.shape[0], w.shape[1]
        ls = np.ascontiguousarray(level_sizes, np.int32)
        assert cells == int((ls.astype(np.int64) ** 2).sum())
        xb, wb = _f16_bits(x), _f16_bits(w)
        bias = np.ascontiguousarray(bias, np.float32)
        y = np.zeros((n, cells, cout), np.uint16)
        self._chk(self.L.yh_op_conv2d_levels_f16(self.h, _p(xb), n, _p(ls), len(ls), cin, _p(wb), _p(bias), cout, k, act, _p(y)))
        return y.view(np.float16).astype(np.float32)

    def op_stem_pool(self, x, w, bias, want_stem=True):
        """x [n][S][S][3], w [64][7][7][3] (f16-representable f32) -> (stem or None, pool) as f32 NHWC."""
        n, S = x.shape[0], x.shape[1]
        so = (S + 6 - 7) // 2 + 1
        po = (so + 2 - 3) // 2 + 1
        xb, wb = _f16_bits(x), _f16_bits(w)
        bias = np.ascontiguousarray(bias, np.float32)
        stem = np.zeros((n, so, so, 64), np.uint16) if want_stem else None
        pool = np.zeros((n, po, po, 64), np.uint16)
        self._chk(self.L.yh_op_stem_pool_f16(self.h, _p(xb), n, S, _p(wb), _p(bias), _p(stem) if want_stem else None, _p(pool)))
        f = lambda a: a.view(np.float16).astype(np.float32)
        return (f(stem) if want_stem else None), f(pool)

    def op_stem_pool_rgb8(self, rgb, w, bias, want_stem=True):
        """rgb [n][S][S][3] uint8 (preprocessing fused into the kernel) -> (stem or None, pool) as f32 NHWC."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        n, S = rgb.shape[0], rgb.shape[1]
        so = (S + 6 - 7) // 2 + 1
        po = (so + 2 - 3) // 2 + 1
        wb = _f16_bits(w)
        bias = np.ascontiguousarray(bias, np.float32)
        stem = np.zeros((n, so, so, 64), np.uint16) if want_stem else None
        pool = np.zeros((n, po, po, 64), np.uint16)
        self._chk(self.L.yh_op_stem_pool_rgb8(self.h, _p(rgb), n, S, _p(wb), _p(bias), _p(stem) if want_stem else None, _p(pool)))
        f = lambda a: a.view(np.float16).astype(np.float32)
        return (f(stem) if want_stem else None), f(pool)

    def op_conv2d_fp8(self, x_codes, w_codes, scale, bias, stride=1, pad=0, residual=None, act=0, reps=0):
        """x_codes [n][h][w][cin] / w_codes [cout][k][k][cin]: uint8 E4M3 codes -> (y f32 NHWC, ms per launch or None)."""
        n, hh, ww, cin = x_codes.shape
        cout, k = w_codes.shape[0], w_codes.shape[1]
        ho, wo = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
        xc, wc = np.ascontiguousarray(x_codes, np.uint8), np.ascontiguousarray(w_codes, np.uint8)
        sc, bs = np.ascontiguousarray(scale, np.float32), np.ascontiguousarray(bias, np.float32)
        rb = _f16_bits(residual) if residual is not None else None
        y = np.zeros((n, ho, wo, cout), np.uint16)
        ms = C.c_float(0)
        self._chk(self.L.yh_op_conv2d_fp8(self.h, _p(xc), n, hh, ww, cin, _p(wc), _p(sc), _p(bs), cout, k, stride, pad,
                                          _p(rb) if rb is not None else None, act, _p(y), reps, C.byref(ms) if reps else None))
        return y.view(np.float16).astype(np.float32), (ms.value if reps else None)

    def op_quantize_e4m3(self, x_f16_bits, inv_scale=1.0):
        """x: uint16 array of f16 bit patterns -> uint8 e4m3 codes of x * inv_scale."""
        xb = np.ascontiguousarray(x_f16_bits, np.uint16)
        y = np.zeros(xb.shape, np.uint8)
        self._chk(self.L.yh_op_quantize_e4m3(self.h, _p(xb), xb.size, C.c_float(inv_scale), _p(y)))
        return y

    def op_detect(self, loc, conf, mask, proto):
        n = loc.shape[0]
        lb, cb, mb, pb = _f16_bits(loc), _f16_bits(conf), _f16_bits(mask), _f16_bits(proto)
        self._chk(self.L.yh_op_detect(self.h, _p(lb), _p(cb), _p(mb), _p(pb), n))
        self.n = n


class Group:
    """RAII wrapper of yh_group: one process, one engine per device (devices may repeat), frames sharded in contiguous blocks."""

    def __init__(self, devices, **engine_kw):
        self.L = load_library()
        cfg = Config()
        self.L.yh_default_config(C.byref(cfg))
        kw = dict(input_size=550, backbone=50, max_batch=1, num_classes=81, top_k=200, max_dets=100, conf_thresh=0.05, nms_thresh=0.5,
                  use_graph=True, debug_tensors=False, precision=PRECISION_F16, tune=None, fp8_f16_layers=0, fp8_per_tensor=False)
        unknown = set(engine_kw) - set(kw)
        if unknown:
            raise TypeError(f"Group: unknown engine keyword(s) {sorted(unknown)}")
        kw.update(engine_kw)
        cfg.backbone, cfg.input_size, cfg.max_batch = kw["backbone"], kw["input_size"], kw["max_batch"]
        cfg.num_classes, cfg.top_k, cfg.max_dets = kw["num_classes"], kw["top_k"], kw["max_dets"]
        cfg.conf_thresh, cfg.nms_thresh, cfg.use_graph = kw["conf_thresh"], kw["nms_thresh"], 1 if kw["use_graph"] else 0
        cfg.debug_tensors, cfg.precision = (1 if kw["debug_tensors"] else 0), kw["precision"]
        cfg.fp8_f16_layers, cfg.fp8_per_tensor = kw["fp8_f16_layers"], (1 if kw["fp8_per_tensor"] else 0)
        if kw["tune"]:
            cfg.tune = Tuning.of(**kw["tune"])
        devs = (C.c_int32 * len(devices))(*devices)
        g = C.c_void_p()
        rc = self.L.yh_group_create(C.byref(cfg), devs, len(devices), C.byref(g))
        if rc != OK:
            raise YhError(rc, self.L.yh_group_last_error(None).decode())
        self.g, self.cfg, self.n, self.S, self.max_batch = g, cfg, len(devices), kw["input_size"], kw["max_batch"]
        self.members = []
        for i in range(self.n):                 # non-owning Engine views of the members (the group owns the handles)
            e = Engine.__new__(Engine)
            e.L, e.cfg, e.h, e._tune = self.L, cfg, C.c_void_p(self.L.yh_group_member(self.g, i)), {}
            e.S, e.C, e.max_batch, e.n = kw["input_size"], kw["num_classes"], kw["max_batch"], 0
            e.P = self.L.yh_num_priors(e.h)
            d = (C.c_int32 * 2)()
            self.L.yh_proto_dims(e.h, C.byref(d))
            e.hp, e.wp = d[0], d[1]
            e.close = lambda: None              # never yh_destroy a member
            self.members.append(e)
        self.hp, self.wp = self.members[0].hp, self.members[0].wp

    def close(self):
        if getattr(self, "g", None):
            for e in self.members:
                e.h = None
            self.L.yh_group_destroy(self.g)
            self.g = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise YhError(rc, self.L.yh_group_last_error(self.g).decode())

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        self._chk(self.L.yh_group_load_weights_host(self.g, _p(blob), blob.size))

    def replicate_weights(self):
        self._chk(self.L.yh_group_replicate_weights(self.g))

    def weights_replication(self):
        return self.L.yh_group_weights_replication(self.g).decode()

    def fp8_calibrate(self):
        self._chk(self.L.yh_group_fp8_calibrate(self.g))

    def evaluate(self, frames, with_tail=True):
        frames = np.ascontiguousarray(frames, np.uint8)
        n = frames.shape[0]
        assert frames.shape == (n, self.S, self.S, 3), frames.shape
        self._chk(self.L.yh_group_evaluate(self.g, _p(frames), n, 1 if with_tail else 0))
        self.total = n

    def prepare(self, n_frames, with_tail=True):
        """yh_group_prepare: capture every member's step for a call with n_frames frames, serially on this thread."""
        self._chk(self.L.yh_group_prepare(self.g, n_frames, 1 if with_tail else 0))

    def evaluate_device(self, dev_ptrs, counts, with_tail=True):
        ptrs = (C.c_void_p * self.n)(*[C.c_void_p(p) for p in dev_ptrs])
        cnt = (C.c_int32 * self.n)(*counts)
        self._chk(self.L.yh_group_evaluate_device(self.g, ptrs, cnt, 1 if with_tail else 0))
        self.total = int(sum(counts))

    def sync(self):
        self._chk(self.L.yh_group_sync(self.g))

    def frame_owner(self, frame):
        m, l = C.c_int32(), C.c_int32()
        self._chk(self.L.yh_group_frame_owner(self.g, frame, C.byref(m), C.byref(l)))
        return m.value, l.value

    def detections(self, frame, want_masks=True):
        nd = C.c_int32()
        dets = (Detection * self.cfg.max_dets)()
        masks = np.zeros((self.cfg.max_dets, self.hp, self.wp), np.uint8) if want_masks else None
        self._chk(self.L.yh_group_read_detections(self.g, frame, C.byref(nd), C.cast(dets, C.c_void_p), self.cfg.max_dets,
                                                  _p(masks) if want_masks else None, masks.size if want_masks else 0))
        out = [dict(class_id=d.class_id, prior=d.prior, score=d.score, box=tuple(d.box)) for d in dets[:nd.value]]
        return out, (masks[:nd.value] if want_masks else None)


class Scene:
    """RAII wrapper of yh_scene: append_scene's two compute dispatches (src/scene.rs:147-331) on the GPU."""

    def __init__(self, width=640, height=480, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.yh_scene_create(device, width, height, C.byref(h))
        if rc != OK:
            raise YhError(rc, self.L.yh_scene_last_error(None).decode())
        self.h, self.W, self.H = h, width, height

    def close(self):
        if getattr(self, "h", None):
            self.L.yh_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise YhError(rc, self.L.yh_scene_last_error(self.h).decode())

    def append(self, depth, cls_id, mode=COMPAT_STRICT):
        depth = np.ascontiguousarray(depth, np.uint16)
        cls_id = np.ascontiguousarray(cls_id, np.uint8)
        assert depth.shape == (self.H, self.W) and cls_id.shape == (self.H, self.W, 2)
        self._chk(self.L.yh_scene_append(self.h, _p(depth), _p(cls_id), mode))

    def append_classified(self, depth, frame_u32=None, frame_dev_ptr=None, mode=COMPAT_STRICT):
        depth = np.ascontiguousarray(depth, np.uint16)
        assert depth.shape == (self.H, self.W)
        if frame_dev_ptr is not None:
            self._chk(self.L.yh_scene_append_classified(self.h, _p(depth), C.c_void_p(frame_dev_ptr), 1, mode))
        else:
            f = np.ascontiguousarray(frame_u32, np.uint32)
            assert f.size == self.W * self.H
            self._chk(self.L.yh_scene_append_classified(self.h, _p(depth), _p(f), 0, mode))

    def read(self):
        out = dict(map=np.zeros((self.H, self.W), np.uint32), world=np.zeros((self.H, self.W, 4), np.float32),
                   conn0=np.zeros((self.H, self.W, 4), np.float32), conn1=np.zeros((self.H, self.W, 4), np.float32),
                   balls=np.zeros((100, 4), np.float32))
        self._chk(self.L.yh_scene_read(self.h, _p(out["map"]), _p(out["world"]), _p(out["conn0"]), _p(out["conn1"]), _p(out["balls"])))
        return out

    def time(self, reps=20):
        ms = C.c_float()
        self._chk(self.L.yh_scene_time(self.h, reps, C.byref(ms)))
        return ms.value


def tfl_validate(model_bytes):
    """Parse-only check of a .tflite buffer (no GPU): returns (ok, n_tensors, n_ops, message)."""
    L = load_library()
    nt, no = C.c_int32(), C.c_int32()
    err = C.create_string_buffer(512)
    rc = L.yh_tfl_validate(model_bytes, len(model_bytes), C.byref(nt), C.byref(no), err, 512)
    return rc == OK, nt.value, no.value, err.value.decode()


_KIND_NP = {1: np.float32, 3: np.uint8, 2: np.int32}


class TfliteEngine:
    """RAII wrapper of yh_tfl: the reference's own model family (uint8 MobileNetV2-style .tflite)."""

    def __init__(self, model_bytes, device=0, tune=None):
        self.L = load_library()
        h = C.c_void_p()
        t = Tuning.of(**(tune or {}))
        rc = self.L.yh_tfl_create_tuned(model_bytes, len(model_bytes), device, C.byref(t), C.byref(h))
        if rc != OK:
            raise YhError(rc, self.L.yh_tfl_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.yh_tfl_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise YhError(rc, self.L.yh_tfl_last_error(self.h).decode())

    @staticmethod
    def _info(ti):
        return dict(name=ti.name.decode(), kind=ti.kind, dims=tuple(ti.dims[:ti.ndims]), scale=ti.scale, zero_point=ti.zero_point)

    def input_info(self):
        ti = TensorInfo()
        self._chk(self.L.yh_tfl_input_info(self.h, C.byref(ti)))
        return self._info(ti)

    def output_count(self):
        return self.L.yh_tfl_output_count(self.h)

    def output_info(self, i):
        ti = TensorInfo()
        self._chk(self.L.yh_tfl_output_info(self.h, i, C.byref(ti)))
        return self._info(ti)

    def set_batch(self, n):
        """1 or 2 images per invoke (the two tiles of a frame as one pass); inputs / outputs / tensors then carry n images."""
        self._chk(self.L.yh_tfl_set_batch(self.h, n))
        self.nb = n

    def set_input(self, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.yh_tfl_set_input(self.h, _p(arr), arr.nbytes))

    def invoke(self):
        self._chk(self.L.yh_tfl_invoke(self.h))

    def output(self, i):
        info = self.output_info(i)
        nb = getattr(self, "nb", 1)
        out = np.empty(((nb,) + tuple(info["dims"])) if nb > 1 else info["dims"], _KIND_NP[info["kind"]])
        self._chk(self.L.yh_tfl_output_read(self.h, i, _p(out), out.nbytes))
        return out

    def plan_summary(self):
        """Launches per invoke of the prepared plan, CONV_2D launches and how many of those run on the int8 matrix pipes."""
        a, b, c = _i(), _i(), _i()
        self._chk(self.L.yh_tfl_plan_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(launches_per_invoke=a.value, conv2d_launches=b.value, conv2d_on_int8_mfma=c.value)

    def tensor(self, index, shape, dtype):
        nb = getattr(self, "nb", 1)
        out = np.empty(((nb,) + tuple(shape)) if nb > 1 else shape, dtype)
        self._chk(self.L.yh_tfl_tensor_read(self.h, index, _p(out), out.nbytes))
        return out

    def classify_frame(self, frame_u32, width, height, mode=COMPAT_STRICT):
        assert frame_u32.dtype == np.uint32 and frame_u32.flags.c_contiguous and frame_u32.size == width * height
        self._chk(self.L.yh_tfl_classify_frame_u32(self.h, _p(frame_u32), width, height, mode))
