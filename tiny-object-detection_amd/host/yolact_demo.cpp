// yolact_demo — the reference's call sequence (src/scene.rs:62-63, :84-93) from compiled host code:
//   yolact_demo <frame_in.u32> <frame_out.u32> <width> <height> <tile> <compat_mode> [model.tflite]
// Reads a packed frame, Yolact::init(), classify in place, writes the frame back. Exit code 6 when
// strict compat reports the reference's non-terminating flood fill (YH_EDIVERGE).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "yolact.hpp"

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s in.u32 out.u32 width height tile compat [model.tflite]\n", argv[0]); return 2; }
    tod::InitOptions opt;
    opt.frame_width = std::atoi(argv[3]);
    opt.frame_height = std::atoi(argv[4]);
    opt.input_size = std::atoi(argv[5]);
    opt.compat_mode = std::atoi(argv[6]);
    if (argc > 7) opt.model_path = argv[7];
    std::vector<std::uint32_t> frame(static_cast<std::size_t>(opt.frame_width) * opt.frame_height);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frame.data(), 4, frame.size(), f) != frame.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    try {
        std::printf("%s\n", tod::Yolact::version().c_str());      // scene.rs:62
        tod::Yolact yolact = tod::Yolact::init(opt);               // scene.rs:63
        yolact.classify(frame.data(), frame.size());               // scene.rs:92
    } catch (const std::runtime_error& e) {
        std::fprintf(stderr, "panicked: %s\n", e.what());
        return std::strstr(e.what(), "does not terminate") ? 6 : 1;
    }
    f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(frame.data(), 4, frame.size(), f) != frame.size()) return 2;
    std::fclose(f);
    // scene.rs:93: the consumer keeps the low 16 bits
    unsigned long long nz = 0;
    for (std::uint32_t px : frame) nz += (px & 0xFFFFu) != 0;
    std::printf("classified %zu px; low-16-bit non-zero: %llu\n", frame.size(), nz);
    return 0;
}
