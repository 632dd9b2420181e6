// Sanitizer harness for the product's TFLite reader + graph validation (csrc/tflite_model.h): parses every file
// named on the command line and prints "ok"/"rejected: <reason>" per file. Built by tests/test_tflite_reader.py
// with -fsanitize=address,undefined: any out-of-bounds read, signed overflow or division by zero aborts it.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "tflite_model.h"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        yh::TflModel m;
        const bool ok = m.parse(b.data(), b.size());
        // what the plan builder computes from a validated file (tflite_exec.hip: prepare / same_pad): must be safe
        long long sink = 0;
        if (ok)
            for (const yh::TflOp& op : m.ops)
                if (op.code == yh::TFL_CONV_2D || op.code == yh::TFL_DEPTHWISE_CONV_2D) {
                    const yh::TflTensor &x = m.tensors[op.in[0]], &w = m.tensors[op.in[1]];
                    const int out_h = (x.shape[1] + op.stride_h - 1) / op.stride_h, eff = (w.shape[1] - 1) * op.dil_h + 1;
                    sink += out_h + eff + (x.shape[2] - ((w.shape[2] - 1) * op.dil_w + 1) + op.stride_w) / op.stride_w;
                    if (op.in.size() > 2 && op.in[2] >= 0) sink += (long long)m.tensors[op.in[2]].count();
                }
        std::printf("%s %s%s [%lld]\n", argv[i], ok ? "ok" : "rejected: ", ok ? "" : m.error.c_str(), sink);
    }
    return 0;
}
