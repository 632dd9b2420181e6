// yolact.hpp — C++ host-side mirror of the reference's `src/yolact.rs` surface over the C ABI
// (include/yolact_hip.h). The reference is compiled code (Rust); Rust is not available in the build
// image, so this is the compiled host side: same items, same argument meaning, same error
// behaviour (the reference `.expect()`s every failure -> here a std::runtime_error carrying the
// same message prefix).
//
//   reference (src/yolact.rs)                      here
//   pub struct Yolact<'a> { interpreter }  :13     class Yolact { yh_engine* | yh_tfl* }
//   pub fn init() -> Yolact<'a>            :17     static Yolact Yolact::init(const InitOptions&)
//   pub(crate) fn classify(&mut self,
//                 frame_buffer: &mut [u32]) :39    void classify(uint32_t* frame_buffer, size_t len)
//   fn classify_tile(frame_buffer, interp) :133    void classify_tile(...)  (interpreter-shaped calls)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/yolact_hip.h"

namespace tod {

struct InitOptions {
    // "data/FRC_model_edgetpu.tflite" is hard-coded at yolact.rs:19; a non-EdgeTPU .tflite path runs
    // through the TFLite model path, an empty path selects the YOLACT engine with synthetic weights.
    std::string model_path;
    int input_size = 224;              // tile size, yolact.rs:143-144
    int frame_width = 640, frame_height = 480;   // yolact.rs:207
    int compat_mode = YH_COMPAT_STRICT;
    int device = 0;
    std::uint64_t seed = 1;
};

class Yolact {
public:
    static Yolact init(const InitOptions& opt = InitOptions());   // yolact.rs:17-37
    Yolact(Yolact&& o) noexcept;
    Yolact& operator=(Yolact&&) = delete;
    Yolact(const Yolact&) = delete;
    ~Yolact();

    // yolact.rs:39-41 + :192-234: overwrites the packed frame (r<<24|g<<16|b<<8, scene.rs:86) in place.
    void classify(std::uint32_t* frame_buffer, std::size_t len);

    // yolact.rs:133-190 with the interpreter calls spelled out (YOLACT engine only): copies one
    // S x S tile in (tensor_data_mut, :161-162), invokes (:163), collects every output as Vec<f32>
    // (:166-188) and returns them; results[4] is what postprocess reads (:91).
    std::vector<std::vector<float>> classify_tile_outputs(const std::uint8_t* rgb_tile);

    static std::string version();      // edgetpu::version(), scene.rs:62

private:
    Yolact() = default;
    yh_engine* engine_ = nullptr;
    yh_tfl* tfl_ = nullptr;
    InitOptions opt_;
};

// One process, N devices: the frame loop of src/main.rs:63-75 / src/scene.rs:77-92 with N frames in flight. A camera batch
// of host frames goes in, the detections of every frame come out; frames shard over the members in contiguous blocks
// (yh_group_*: one host worker thread + stream per device, weights replicated once, no per-step collective).
struct GroupOptions {
    std::vector<int> devices = {0};    // one member per entry; an entry may repeat (two members then share that device)
    int input_size = 550, backbone = YH_BACKBONE_R50;
    int frames_per_member = 8;         // yh_config.max_batch of every member
    int precision = YH_PRECISION_F16;
    std::uint64_t seed = 1;            // seeded synthetic weights (the reference's model file is absent)
};

struct FrameDetections {
    std::vector<yh_detection> dets;
    std::vector<std::uint8_t> masks;   // dets.size() x Hp x Wp, 0/1 (empty unless asked for)
};

class YolactGroup {
public:
    static YolactGroup init(const GroupOptions& opt = GroupOptions());
    YolactGroup(YolactGroup&& o) noexcept;
    YolactGroup& operator=(YolactGroup&&) = delete;
    YolactGroup(const YolactGroup&) = delete;
    ~YolactGroup();

    int members() const;
    int capacity() const { return members() * opt_.frames_per_member; }   // frames per evaluate
    std::string weights_replication() const;
    // captures every member's step for calls with n frames (init() does it for capacity()); serial, on the calling thread
    void prepare(int n_frames);
    // n frames of u8 RGB [n][S][S][3]: enqueues them on the members and returns (the GPUs run on; the frames are free again)
    void evaluate(const std::uint8_t* frames, int n);
    // waits for the member that holds `frame` of the last evaluate and returns its detections
    FrameDetections detections(int frame, bool want_masks = false);
    void sync();

private:
    YolactGroup() = default;
    yh_group* g_ = nullptr;
    GroupOptions opt_;
    int hp_ = 0, wp_ = 0, max_dets_ = 100;
};

}  // namespace tod
