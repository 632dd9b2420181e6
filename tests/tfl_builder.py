"""Minimal TensorFlow Lite flatbuffer WRITER for tests: builds synthetic .tflite models (schema v3,
file identifier "TFL3") so the product's reader + executor can be checked without any TFLite
runtime (none is installable here) and without the reference's absent FRC_model.tflite.

Layout strategy: every object is placed AFTER its parent (all uoffsets point forward), each table's
vtable directly in front of it. Only the schema subset the reader understands is emitted.
"""
import struct

import numpy as np

BUILTIN = {"ADD": 0, "CONCATENATION": 2, "CONV_2D": 3, "DEPTHWISE_CONV_2D": 4, "DEQUANTIZE": 6, "RELU": 19,
           "RESHAPE": 22, "RESIZE_BILINEAR": 23, "TANH": 28, "PAD": 34, "QUANTIZE": 114}
OPTIONS_TYPE = {"CONV_2D": 1, "DEPTHWISE_CONV_2D": 2, "CONCATENATION": 10, "ADD": 11, "RESIZE_BILINEAR": 15,
                "RESHAPE": 17, "PAD": 22, "DEQUANTIZE": 61, "QUANTIZE": 87, "TANH": 0, "RELU": 0}
TENSOR_TYPE = {"f32": 0, "i32": 2, "u8": 3}
NP_TYPE = {"f32": np.float32, "i32": np.int32, "u8": np.uint8}


class T:
    def __init__(self, name, shape, dtype="u8", data=None, scale=None, zp=0):
        self.name, self.shape, self.dtype, self.scale, self.zp = name, tuple(int(s) for s in shape), dtype, scale, int(zp)
        self.data = None if data is None else np.ascontiguousarray(data, NP_TYPE[dtype]).reshape(self.shape)


class Op:
    def __init__(self, code, inputs, outputs, **opts):
        self.code, self.inputs, self.outputs, self.opts = code, list(inputs), list(outputs), opts


class Model:
    def __init__(self):
        self.tensors, self.ops, self.inputs, self.outputs = [], [], [], []

    def add(self, t):
        self.tensors.append(t)
        return len(self.tensors) - 1


# ------------------------------------------------------------------ flatbuffer objects
class _Obj:
    pos = None


class _Str(_Obj):
    def __init__(self, s):
        self.b = s.encode()

    def size(self):
        return 4 + len(self.b) + 1

    def align(self):
        return 4

    def children(self):
        return []

    def emit(self, buf):
        struct.pack_into("<I", buf, self.pos, len(self.b))
        buf[self.pos + 4:self.pos + 4 + len(self.b)] = self.b


class _Vec(_Obj):
    """kind: struct format char ('b','B','i','I','f','q') or 'off' (vector of offsets to objects)."""

    def __init__(self, kind, items, data_align=4):
        self.kind, self.items, self.da = kind, items, data_align

    def esize(self):
        return 4 if self.kind == "off" else struct.calcsize("<" + self.kind)

    def size(self):
        return self._data_off() + self.esize() * len(self.items)

    def _data_off(self):
        # the element block starts data_align-aligned relative to pos (pos itself is aligned to max(4, da))
        return 4 if self.da <= 4 else self.da

    def align(self):
        return max(4, self.da, self.esize())

    def children(self):
        return self.items if self.kind == "off" else []

    def emit(self, buf):
        # length sits directly in front of the elements
        d = self.pos + self._data_off()
        struct.pack_into("<I", buf, d - 4, len(self.items))
        self.len_pos = d - 4
        for i, it in enumerate(self.items):
            if self.kind == "off":
                struct.pack_into("<I", buf, d + 4 * i, it.pos_ref() - (d + 4 * i))
            else:
                struct.pack_into("<" + self.kind, buf, d + self.esize() * i, it)

    def pos_ref(self):
        return self.pos + self._data_off() - 4


def _ref(o):
    return o.pos_ref() if isinstance(o, _Vec) else (o.tpos if isinstance(o, _Table) else o.pos)


_Str.pos_ref = lambda self: self.pos


class _Table(_Obj):
    """fields: {id: (fmt, value)} with fmt a struct char for scalars or 'off' for a child object."""

    def __init__(self, fields):
        self.fields = {k: v for k, v in fields.items() if v is not None and v[1] is not None}
        self.nf = (max(self.fields) + 1) if self.fields else 0
        off, self.foff = 4, {}
        for fid in sorted(self.fields, key=lambda f: -self._fsize(f)):   # big fields first: natural alignment
            sz = self._fsize(fid)
            off = (off + sz - 1) // sz * sz
            self.foff[fid] = off
            off += sz
        self.tsize = (off + 3) // 4 * 4
        self.vsize = 4 + 2 * self.nf
        self.vpad = (-self.vsize) % 4

    def _fsize(self, fid):
        f = self.fields[fid][0]
        return 4 if f == "off" else struct.calcsize("<" + f)

    def size(self):
        return self.vsize + self.vpad + self.tsize

    def align(self):
        return 8

    def children(self):
        return [v for f, v in self.fields.values() if f == "off"]

    def pos_ref(self):
        return self.tpos

    def emit(self, buf):
        v = self.pos
        self.tpos = t = self.pos + self.vsize + self.vpad
        struct.pack_into("<HH", buf, v, self.vsize, self.tsize)
        for fid in range(self.nf):
            struct.pack_into("<H", buf, v + 4 + 2 * fid, self.foff.get(fid, 0))
        struct.pack_into("<i", buf, t, t - v)
        for fid, (f, val) in self.fields.items():
            p = t + self.foff[fid]
            if f == "off":
                struct.pack_into("<I", buf, p, val.pos_ref() - p)
            else:
                struct.pack_into("<" + f, buf, p, val)


def _layout(root):
    order, pos, stack = [], 8, [root]
    seen = set()
    while stack:   # parents before children, every object once
        o = stack.pop(0)
        if id(o) in seen:
            continue
        seen.add(id(o))
        a = o.align()
        pos = (pos + a - 1) // a * a
        o.pos = pos
        if isinstance(o, _Table):
            o.tpos = pos + o.vsize + o.vpad
        pos += o.size()
        order.append(o)
        stack.extend(o.children())
    return order, pos


def _options(op):
    o, c = op.opts, op.code
    if c == "CONV_2D":
        return _Table({0: ("b", o["padding"]), 1: ("i", o["stride_w"]), 2: ("i", o["stride_h"]), 3: ("b", o.get("act", 0)),
                       4: ("i", o.get("dil_w", 1)), 5: ("i", o.get("dil_h", 1))})
    if c == "DEPTHWISE_CONV_2D":
        return _Table({0: ("b", o["padding"]), 1: ("i", o["stride_w"]), 2: ("i", o["stride_h"]), 3: ("i", o.get("depth_multiplier", 1)),
                       4: ("b", o.get("act", 0)), 5: ("i", o.get("dil_w", 1)), 6: ("i", o.get("dil_h", 1))})
    if c == "ADD":
        return _Table({0: ("b", o.get("act", 0))})
    if c == "CONCATENATION":
        return _Table({0: ("i", o["axis"]), 1: ("b", 0)})
    if c == "RESIZE_BILINEAR":
        return _Table({2: ("B", 1 if o.get("align_corners") else 0), 3: ("B", 1 if o.get("half_pixel_centers") else 0)})
    if c == "RESHAPE":
        return _Table({0: ("off", _Vec("i", [int(v) for v in o["new_shape"]]))})
    if c in ("PAD", "QUANTIZE", "DEQUANTIZE"):
        return _Table({})
    return None


def serialize(model):
    codes = sorted({op.code for op in model.ops}, key=lambda c: BUILTIN[c])
    buffers = [_Table({})]   # buffer 0: the empty sentinel
    tensors = []
    for t in model.tensors:
        bidx = 0
        if t.data is not None:
            buffers.append(_Table({0: ("off", _Vec("B", list(t.data.tobytes()), data_align=16))}))
            bidx = len(buffers) - 1
        q = None
        if t.scale is not None:
            q = _Table({2: ("off", _Vec("f", [float(t.scale)])), 3: ("off", _Vec("q", [int(t.zp)]))})
        tensors.append(_Table({0: ("off", _Vec("i", list(t.shape))), 1: ("b", TENSOR_TYPE[t.dtype]), 2: ("I", bidx),
                               3: ("off", _Str(t.name)), 4: ("off", q) if q else None}))
    ops = []
    for op in model.ops:
        opt = _options(op)
        ops.append(_Table({0: ("I", codes.index(op.code)), 1: ("off", _Vec("i", op.inputs)), 2: ("off", _Vec("i", op.outputs)),
                           3: ("B", OPTIONS_TYPE[op.code]) if opt is not None else None, 4: ("off", opt) if opt is not None else None}))
    sub = _Table({0: ("off", _Vec("off", tensors)), 1: ("off", _Vec("i", model.inputs)), 2: ("off", _Vec("i", model.outputs)),
                  3: ("off", _Vec("off", ops)), 4: ("off", _Str("main"))})
    opcodes = [_Table({0: ("b", min(BUILTIN[c], 127)), 2: ("i", 1), 3: ("i", BUILTIN[c])}) for c in codes]
    root = _Table({0: ("I", 3), 1: ("off", _Vec("off", opcodes)), 2: ("off", _Vec("off", [sub])),
                   3: ("off", _Str("yolact_hip synthetic test model")), 4: ("off", _Vec("off", buffers))})
    order, end = _layout(root)
    buf = bytearray(end)
    for o in order:
        o.emit(buf)
    struct.pack_into("<I", buf, 0, root.tpos)
    buf[4:8] = b"TFL3"
    return bytes(buf)
