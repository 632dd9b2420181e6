// tflite_model.h — dependency-free, bounds-checked reader of TensorFlow Lite flatbuffers (schema v3).
//
// Stands in for FlatBufferModel::build_from_file + the interpreter's graph walk
// (/root/reference/src/yolact.rs:18-35): tensors (shape, type, constant data, per-tensor
// quantisation), operators with the builtin options of the op set listed in
// data/FRC_model_edgetpu.log:7-19, subgraph inputs/outputs. Every offset is range-checked: a
// malformed file yields an error string, never an out-of-bounds read.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace yh {

enum TflType { TFL_F32 = 0, TFL_I32 = 2, TFL_U8 = 3, TFL_I64 = 4, TFL_I8 = 9 };
enum TflOpCode { TFL_ADD = 0, TFL_CONCATENATION = 2, TFL_CONV_2D = 3, TFL_DEPTHWISE_CONV_2D = 4, TFL_DEQUANTIZE = 6,
                 TFL_RELU = 19, TFL_RELU6 = 21, TFL_RESHAPE = 22, TFL_RESIZE_BILINEAR = 23, TFL_TANH = 28, TFL_PAD = 34,
                 TFL_QUANTIZE = 114, TFL_CUSTOM = 32 };

struct TflTensor {
    std::vector<int> shape;
    int type = 0;
    const uint8_t* data = nullptr;  // constant data inside the model buffer (nullptr: activation)
    size_t nbytes = 0;
    std::string name;
    bool quant = false;
    float scale = 1.0f;
    int zp = 0;
    size_t count() const { size_t c = 1; for (int s : shape) c *= (size_t)s; return c; }   // (parse() bounds it: <= kMaxElems)
    static constexpr uint64_t kMaxElems = 1ull << 31;
    // element count with overflow checking: false if a dimension is negative or the product exceeds kMaxElems
    bool count_checked(uint64_t* out) const {
        uint64_t c = 1;
        for (int s : shape) {
            if (s < 0) return false;
            if (s != 0 && c > kMaxElems / (uint64_t)s) return false;
            c *= (uint64_t)s;
        }
        *out = c;
        return true;
    }
    size_t elem() const { return type == TFL_U8 || type == TFL_I8 ? 1 : (type == TFL_I64 ? 8 : 4); }
};

struct TflOp {
    int code = -1;
    std::string custom;
    std::vector<int> in, out;
    int padding = 0, stride_w = 1, stride_h = 1, act = 0, depth_mult = 1, dil_w = 1, dil_h = 1, axis = 0;
    bool align_corners = false, half_pixel = false;
};

class TflModel {
public:
    std::vector<TflTensor> tensors;
    std::vector<TflOp> ops;
    std::vector<int> inputs, outputs;
    std::string error;

    bool parse(const uint8_t* p, size_t n) {
        b_ = p; n_ = n; ok_ = true; error.clear();
        if (n < 8) return fail("file too small");
        const uint32_t root = rd32(0);
        if (n >= 8 && memcmp(p + 4, "TFL3", 4) != 0) return fail("missing TFL3 file identifier");
        if (!table(root)) return fail("bad root table");
        const uint32_t opcodes = vec(root, 1), subgraphs = vec(root, 2), buffers = vec(root, 4);
        if (!ok_ || !subgraphs || vlen(subgraphs) < 1) return fail("model has no subgraph");
        std::vector<int> codes;
        std::vector<std::string> customs;
        // (a forged vector length must not drive the loop: every entry needs 4 bytes of file)
        if (opcodes && (uint64_t)vlen(opcodes) > n_ / 4) return fail("operator-code vector longer than the file");
        for (uint32_t i = 0; ok_ && i < vlen(opcodes); ++i) {
            const uint32_t oc = vtab(opcodes, i);
            int dep = (int8_t)scalar8(oc, 0, 0), full = (int)scalar32(oc, 3, 0);
            codes.push_back(full > dep ? full : dep);
            customs.push_back(str(oc, 1));
        }
        const uint32_t sg = vtab(subgraphs, 0);
        const uint32_t tv = vec(sg, 0), ov = vec(sg, 3);
        ints(vec(sg, 1), &inputs);
        ints(vec(sg, 2), &outputs);
        for (uint32_t i = 0; ok_ && i < vlen(tv); ++i) {
            const uint32_t t = vtab(tv, i);
            TflTensor x;
            ints(vec(t, 0), &x.shape);
            x.type = (int8_t)scalar8(t, 1, 0);
            const uint32_t bi = scalar32(t, 2, 0);
            x.name = str(t, 3);
            const uint32_t q = sub(t, 4);
            if (q) {
                const uint32_t sv = vec(q, 2), zv = vec(q, 3);
                if (sv && vlen(sv) >= 1) {
                    x.quant = true;
                    uint32_t bits = rd32(sv + 4);
                    memcpy(&x.scale, &bits, 4);
                    if (vlen(sv) > 1) return fail("per-channel quantisation is not supported (tensor " + x.name + ")");
                }
                if (zv && vlen(zv) >= 1) x.zp = (int)(int64_t)rd64(zv + 4);
            }
            if (bi != 0) {
                if (!buffers || bi >= vlen(buffers)) return fail("tensor buffer index out of range");
                const uint32_t bt = vtab(buffers, bi), dv = vec(bt, 0);
                if (dv && vlen(dv) > 0) {
                    if (!range(dv + 4, vlen(dv))) return fail("buffer data out of range");
                    x.data = b_ + dv + 4;
                    x.nbytes = vlen(dv);
                }
            }
            for (int s : x.shape) if (s < 0 || s > (1 << 24)) return fail("unsupported tensor shape in " + x.name);
            uint64_t ne = 0;
            if (x.shape.size() > 8 || !x.count_checked(&ne)) return fail("tensor too large or of too high a rank: " + x.name);
            if (x.data && x.nbytes != x.count() * x.elem()) return fail("constant size mismatch in " + x.name);
            tensors.push_back(x);
        }
        for (uint32_t i = 0; ok_ && i < vlen(ov); ++i) {
            const uint32_t o = vtab(ov, i);
            TflOp op;
            const uint32_t ci = scalar32(o, 0, 0);
            if (ci >= codes.size()) return fail("opcode index out of range");
            op.code = codes[ci];
            op.custom = customs[ci];
            ints(vec(o, 1), &op.in);
            ints(vec(o, 2), &op.out);
            const uint32_t opt = sub(o, 4);
            if (opt) {
                switch (op.code) {
                    case TFL_CONV_2D:
                        op.padding = (int8_t)scalar8(opt, 0, 0); op.stride_w = (int)scalar32(opt, 1, 1); op.stride_h = (int)scalar32(opt, 2, 1);
                        op.act = (int8_t)scalar8(opt, 3, 0); op.dil_w = (int)scalar32(opt, 4, 1); op.dil_h = (int)scalar32(opt, 5, 1);
                        break;
                    case TFL_DEPTHWISE_CONV_2D:
                        op.padding = (int8_t)scalar8(opt, 0, 0); op.stride_w = (int)scalar32(opt, 1, 1); op.stride_h = (int)scalar32(opt, 2, 1);
                        op.depth_mult = (int)scalar32(opt, 3, 1); op.act = (int8_t)scalar8(opt, 4, 0);
                        op.dil_w = (int)scalar32(opt, 5, 1); op.dil_h = (int)scalar32(opt, 6, 1);
                        break;
                    case TFL_ADD: op.act = (int8_t)scalar8(opt, 0, 0); break;
                    case TFL_CONCATENATION: op.axis = (int)scalar32(opt, 0, 0); op.act = (int8_t)scalar8(opt, 1, 0); break;
                    case TFL_RESIZE_BILINEAR: op.align_corners = scalar8(opt, 2, 0) != 0; op.half_pixel = scalar8(opt, 3, 0) != 0; break;
                    default: break;
                }
            }
            for (int t : op.in) if (t >= (int)tensors.size()) return fail("operator input index out of range");
            for (int t : op.out) if (t < 0 || t >= (int)tensors.size()) return fail("operator output index out of range");
            ops.push_back(op);
        }
        for (int t : inputs) if (t < 0 || t >= (int)tensors.size()) return fail("graph input out of range");
        for (int t : outputs) if (t < 0 || t >= (int)tensors.size()) return fail("graph output out of range");
        if (!ok_) return fail("truncated or malformed flatbuffer");
        return validate_graph();
    }

    // Everything the executor's plan builder (tflite_exec.hip: prepare) dereferences, checked WITHOUT a GPU: operator
    // arity, every tensor index it reads (only a convolution's bias may be the optional -1), ranks, kernel / stride /
    // dilation / depth-multiplier >= 1 (they are divisors), constant operands present where the plan reads them on the
    // host. A file that passes cannot make prepare() index out of range or divide by zero; what is merely unsupported
    // (another dtype, a dynamic shape) is still reported by prepare() with its own message.
    bool validate_graph() {
        const int nt = (int)tensors.size();
        auto idx = [&](int t) { return t >= 0 && t < nt; };
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            const TflOp& op = ops[oi];
            const std::string at = " (operator " + std::to_string(oi) + ")";
            for (int t : op.out) if (!idx(t)) return fail("operator output index out of range" + at);
            auto ins = [&](size_t lo, size_t hi) { return op.in.size() >= lo && op.in.size() <= hi; };
            auto all_in = [&]() { for (int t : op.in) if (!idx(t)) return false; return true; };
            auto rank = [&](int t, size_t r) { return tensors[t].shape.size() == r; };
            switch (op.code) {
                case TFL_CONV_2D:
                case TFL_DEPTHWISE_CONV_2D: {
                    if (!ins(2, 3) || op.out.size() != 1 || !idx(op.in[0]) || !idx(op.in[1])) return fail("conv: bad arity or operand index" + at);
                    if (op.in.size() == 3 && op.in[2] != -1 && !idx(op.in[2])) return fail("conv: bias index out of range" + at);
                    if (!rank(op.in[0], 4) || !rank(op.in[1], 4) || !rank(op.out[0], 4)) return fail("conv: operands must be 4-D" + at);
                    if (op.stride_h < 1 || op.stride_w < 1 || op.dil_h < 1 || op.dil_w < 1 || op.depth_mult < 1 ||
                        op.stride_h > 64 || op.stride_w > 64 || op.dil_h > 64 || op.dil_w > 64)
                        return fail("conv: stride, dilation and depth multiplier must be in 1..64" + at);
                    const std::vector<int>& w = tensors[op.in[1]].shape;
                    if (w[1] < 1 || w[2] < 1 || w[0] < 1 || w[3] < 1) return fail("conv: empty kernel" + at);
                    for (int d : tensors[op.in[0]].shape) if (d < 1) return fail("conv: empty input" + at);
                    break;
                }
                case TFL_ADD:
                    if (!ins(2, 2) || op.out.size() != 1 || !all_in()) return fail("add: bad arity or operand index" + at);
                    break;
                case TFL_RELU: case TFL_RELU6: case TFL_QUANTIZE: case TFL_DEQUANTIZE: case TFL_TANH:
                    if (!ins(1, 1) || op.out.size() != 1 || !all_in()) return fail("unary operator: bad arity or operand index" + at);
                    break;
                case TFL_PAD:
                    if (!ins(2, 2) || op.out.size() != 1 || !all_in()) return fail("pad: bad arity or operand index" + at);
                    if (!rank(op.in[0], 4) || !rank(op.out[0], 4)) return fail("pad: input and output must be 4-D" + at);
                    if (tensors[op.in[1]].type != TFL_I32 || !tensors[op.in[1]].data || tensors[op.in[1]].nbytes != 32) return fail("pad: paddings must be a constant int32 [4,2]" + at);
                    break;
                case TFL_RESIZE_BILINEAR:
                    if (!ins(2, 2) || op.out.size() != 1 || !all_in()) return fail("resize_bilinear: bad arity or operand index" + at);
                    if (!rank(op.in[0], 4) || !rank(op.out[0], 4)) return fail("resize_bilinear: input and output must be 4-D" + at);
                    if (tensors[op.in[1]].type != TFL_I32 || !tensors[op.in[1]].data || tensors[op.in[1]].nbytes != 8) return fail("resize_bilinear: size must be a constant int32 [2]" + at);
                    for (int d : tensors[op.in[0]].shape) if (d < 1) return fail("resize_bilinear: empty input" + at);
                    for (int d : tensors[op.out[0]].shape) if (d < 1) return fail("resize_bilinear: empty output" + at);
                    break;
                case TFL_CONCATENATION: {
                    if (op.in.empty() || op.out.size() != 1 || !all_in()) return fail("concatenation: bad arity or operand index" + at);
                    const size_t nd = tensors[op.out[0]].shape.size();
                    if (nd < 1) return fail("concatenation: scalar output" + at);
                    for (int t : op.in) if (tensors[t].shape.size() != nd) return fail("concatenation: rank mismatch" + at);
                    break;
                }
                case TFL_RESHAPE:
                    if (!ins(1, 2) || op.out.size() != 1 || !idx(op.in[0])) return fail("reshape: bad arity or operand index" + at);
                    break;
                default:
                    break;   // unsupported / custom operators: prepare() refuses them by name before touching any operand
            }
        }
        return true;
    }

private:
    const uint8_t* b_ = nullptr;
    size_t n_ = 0;
    bool ok_ = true;

    bool fail(const std::string& m) { if (error.empty()) error = m; ok_ = false; return false; }
    bool range(uint64_t off, uint64_t len) const { return off <= n_ && len <= n_ - off; }
    uint16_t rd16(uint32_t o) { uint16_t v = 0; if (range(o, 2)) memcpy(&v, b_ + o, 2); else ok_ = false; return v; }
    uint32_t rd32(uint32_t o) { uint32_t v = 0; if (range(o, 4)) memcpy(&v, b_ + o, 4); else ok_ = false; return v; }
    uint64_t rd64(uint32_t o) { uint64_t v = 0; if (range(o, 8)) memcpy(&v, b_ + o, 8); else ok_ = false; return v; }
    bool table(uint32_t t) { return range(t, 4); }
    // absolute position of field `id` of table t, or 0 when absent
    uint32_t field(uint32_t t, int id) {
        if (!t) return 0;
        const int32_t so = (int32_t)rd32(t);
        const int64_t vt = (int64_t)t - so;
        if (vt < 0 || !range((uint64_t)vt, 4)) { ok_ = false; return 0; }
        const uint16_t vsz = rd16((uint32_t)vt);
        if (4 + 2 * id + 2 > vsz) return 0;
        const uint16_t fo = rd16((uint32_t)vt + 4 + 2 * id);
        return fo ? t + fo : 0;
    }
    uint32_t scalar32(uint32_t t, int id, uint32_t def) { const uint32_t f = field(t, id); return f ? rd32(f) : def; }
    uint8_t scalar8(uint32_t t, int id, uint8_t def) { const uint32_t f = field(t, id); if (!f) return def; if (!range(f, 1)) { ok_ = false; return def; } return b_[f]; }
    uint32_t indirect(uint32_t t, int id) { const uint32_t f = field(t, id); if (!f) return 0; const uint32_t tgt = f + rd32(f); if (!range(tgt, 4)) { ok_ = false; return 0; } return tgt; }
    uint32_t vec(uint32_t t, int id) { return indirect(t, id); }   // position of the length word
    uint32_t sub(uint32_t t, int id) { return indirect(t, id); }
    uint32_t vlen(uint32_t v) { return v ? rd32(v) : 0; }
    uint32_t vtab(uint32_t v, uint32_t i) {  // i-th table of a vector of offsets
        const uint32_t e = v + 4 + 4 * i;
        const uint32_t tgt = e + rd32(e);
        if (!range(tgt, 4)) { ok_ = false; return 0; }
        return tgt;
    }
    std::string str(uint32_t t, int id) {
        const uint32_t s = indirect(t, id);
        if (!s) return std::string();
        const uint32_t len = rd32(s);
        if (!range(s + 4, len)) { ok_ = false; return std::string(); }
        return std::string((const char*)b_ + s + 4, len);
    }
    void ints(uint32_t v, std::vector<int>* out) {
        out->clear();
        if (!v) return;
        const uint32_t len = vlen(v);
        if (!range((uint64_t)v + 4, (uint64_t)len * 4)) { ok_ = false; return; }
        for (uint32_t i = 0; i < len; ++i) out->push_back((int)rd32(v + 4 + 4 * i));
    }
};

}  // namespace yh
