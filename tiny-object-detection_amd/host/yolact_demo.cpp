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

// yolact_demo --group <frames.u8> <out.txt> <n_frames> <size> <frames_per_member> <dev[,dev...]>
// The sharded frame loop from compiled host code: n frames [n][S][S][3] u8 go through tod::YolactGroup (one member per listed
// device, contiguous blocks), every frame's detections are written as text: "frame class prior score_bits x1 y1 x2 y2 (bits)".
static int group_main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s --group frames.u8 out.txt n_frames size frames_per_member dev[,dev...]\n", argv[0]); return 2; }
    const int n = std::atoi(argv[4]);
    tod::GroupOptions opt;
    opt.input_size = std::atoi(argv[5]);
    opt.frames_per_member = std::atoi(argv[6]);
    opt.devices.clear();
    for (const char* q = argv[7]; *q;) { opt.devices.push_back(std::atoi(q)); const char* c = std::strchr(q, ','); if (!c) break; q = c + 1; }
    std::vector<std::uint8_t> frames(static_cast<std::size_t>(n) * opt.input_size * opt.input_size * 3);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    std::fclose(f);
    try {
        tod::YolactGroup group = tod::YolactGroup::init(opt);
        std::printf("%s\n%d members; weights: %s\n", tod::Yolact::version().c_str(), group.members(), group.weights_replication().c_str());
        group.evaluate(frames.data(), n);
        f = std::fopen(argv[3], "w");
        if (!f) return 2;
        for (int i = 0; i < n; ++i) {
            const tod::FrameDetections d = group.detections(i);
            for (const yh_detection& x : d.dets) {
                std::uint32_t b[5];
                std::memcpy(&b[0], &x.score, 4);
                std::memcpy(&b[1], x.box, 16);
                std::fprintf(f, "%d %d %d %08x %08x %08x %08x %08x\n", i, x.class_id, x.prior, b[0], b[1], b[2], b[3], b[4]);
            }
        }
        std::fclose(f);
    } catch (const std::runtime_error& e) {
        std::fprintf(stderr, "panicked: %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "--group") == 0) return group_main(argc, argv);
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
