// yolact.cpp — see yolact.hpp. Thin, allocation-free per call; all arithmetic is in libyolact_hip.so.
#include "yolact.hpp"

#include <fstream>
#include <iterator>
#include <stdexcept>
#include <utility>

namespace tod {

namespace {
[[noreturn]] void expect_failed(const char* what, const char* detail) {   // Rust: .expect("what")
    throw std::runtime_error(std::string(what) + ": " + (detail ? detail : ""));
}
}  // namespace

std::string Yolact::version() { return yh_version(); }

Yolact Yolact::init(const InitOptions& opt) {
    Yolact y;
    y.opt_ = opt;
    if (!opt.model_path.empty()) {
        std::ifstream f(opt.model_path, std::ios::binary);
        if (!f) expect_failed("failed to load model", opt.model_path.c_str());            // yolact.rs:20
        std::vector<char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (yh_tfl_create(bytes.data(), bytes.size(), opt.device, &y.tfl_) != YH_OK)
            expect_failed("must build interpreter", yh_tfl_last_error(nullptr));           // yolact.rs:29
        return y;
    }
    yh_config cfg;
    yh_default_config(&cfg);
    cfg.device = opt.device;
    cfg.input_size = opt.input_size;
    cfg.max_batch = 2;                         // the two tiles of yolact.rs:216-217 run as one batch
    if (yh_create(&cfg, &y.engine_) != YH_OK) expect_failed("must create interpreter builder", yh_last_error(nullptr));   // :25
    const std::size_t n = yh_weights_nbytes(y.engine_);
    std::vector<unsigned char> blob(n);
    if (yh_weights_generate(y.engine_, opt.seed, blob.data(), n) != YH_OK || yh_load_weights_host(y.engine_, blob.data(), n) != YH_OK)
        expect_failed("failed to allocate tensors.", yh_last_error(y.engine_));            // yolact.rs:35
    return y;
}

Yolact::Yolact(Yolact&& o) noexcept : engine_(o.engine_), tfl_(o.tfl_), opt_(std::move(o.opt_)) { o.engine_ = nullptr; o.tfl_ = nullptr; }

Yolact::~Yolact() {
    if (engine_) yh_destroy(engine_);
    if (tfl_) yh_tfl_destroy(tfl_);
}

void Yolact::classify(std::uint32_t* frame_buffer, std::size_t len) {
    if (len != static_cast<std::size_t>(opt_.frame_width) * opt_.frame_height)
        expect_failed("frame buffer", "length does not match width * height");            // ImageBuffer::from_vec(..).unwrap(), :207
    const int rc = engine_ ? yh_classify_frame_u32(engine_, frame_buffer, opt_.frame_width, opt_.frame_height, opt_.compat_mode)
                           : yh_tfl_classify_frame_u32(tfl_, frame_buffer, opt_.frame_width, opt_.frame_height, opt_.compat_mode);
    if (rc != YH_OK) expect_failed("invoke failed", engine_ ? yh_last_error(engine_) : yh_tfl_last_error(tfl_));   // :163
}

std::vector<std::vector<float>> Yolact::classify_tile_outputs(const std::uint8_t* rgb_tile) {
    if (!engine_) expect_failed("classify_tile", "interpreter-shaped calls are exposed for the YOLACT engine");
    std::int32_t dims[4];
    yh_input_dims(engine_, dims);                                                          // :149-150
    if (yh_set_input_u8(engine_, rgb_tile, 1) != YH_OK) expect_failed("must data", yh_last_error(engine_));   // :161-162
    if (yh_invoke(engine_) != YH_OK) expect_failed("invoke failed", yh_last_error(engine_));                    // :163
    std::vector<std::vector<float>> results;
    for (int i = 0; i < yh_output_count(engine_); ++i) {                                   // :166-188
        yh_tensor_info info;
        if (yh_output_info(engine_, i, &info) != YH_OK) expect_failed("must data", yh_last_error(engine_));
        std::size_t n = 1;
        for (int d = 0; d < info.ndims; ++d) n *= static_cast<std::size_t>(info.dims[d]);
        std::vector<float> out(n);
        if (yh_output_read_f32(engine_, i, out.data(), n) != YH_OK) expect_failed("must data", yh_last_error(engine_));
        results.push_back(std::move(out));
    }
    return results;
}

// ---- YolactGroup ---------------------------------------------------------------------------------------------------
YolactGroup YolactGroup::init(const GroupOptions& opt) {
    YolactGroup y;
    y.opt_ = opt;
    yh_config cfg;
    yh_default_config(&cfg);
    cfg.input_size = opt.input_size;
    cfg.backbone = opt.backbone;
    cfg.max_batch = opt.frames_per_member;
    cfg.precision = opt.precision;
    y.max_dets_ = cfg.max_dets;
    std::vector<std::int32_t> devs(opt.devices.begin(), opt.devices.end());
    if (yh_group_create(&cfg, devs.data(), static_cast<std::int32_t>(devs.size()), &y.g_) != YH_OK)
        expect_failed("must create interpreter builder", yh_group_last_error(nullptr));                     // yolact.rs:25, once per device
    yh_engine* m0 = yh_group_member(y.g_, 0);
    const std::size_t n = yh_weights_nbytes(m0);
    std::vector<unsigned char> blob(n);
    if (yh_weights_generate(m0, opt.seed, blob.data(), n) != YH_OK) expect_failed("failed to load model", yh_last_error(m0));   // :20
    if (yh_group_load_weights_host(y.g_, blob.data(), n) != YH_OK) expect_failed("failed to allocate tensors.", yh_group_last_error(y.g_));   // :35
    std::int32_t pd[2] = {0, 0};
    yh_proto_dims(m0, pd);
    y.hp_ = pd[0]; y.wp_ = pd[1];
    // allocate_tensors() (yolact.rs:35) for the full-capacity call: every member's step is captured now, on this thread, one member
    // after the other - no capture is left for the frame loop (fp8 precision: after the calibration instead, see prepare())
    if (opt.precision == YH_PRECISION_F16) y.prepare(y.capacity());
    return y;
}

void YolactGroup::prepare(int n_frames) {
    if (yh_group_prepare(g_, n_frames, 1) != YH_OK) expect_failed("failed to allocate tensors.", yh_group_last_error(g_));   // yolact.rs:35
}

YolactGroup::YolactGroup(YolactGroup&& o) noexcept : g_(o.g_), opt_(std::move(o.opt_)), hp_(o.hp_), wp_(o.wp_), max_dets_(o.max_dets_) { o.g_ = nullptr; }
YolactGroup::~YolactGroup() { if (g_) yh_group_destroy(g_); }
int YolactGroup::members() const { return yh_group_size(g_); }
std::string YolactGroup::weights_replication() const { return yh_group_weights_replication(g_); }

void YolactGroup::evaluate(const std::uint8_t* frames, int n) {
    if (yh_group_evaluate(g_, frames, n, 1) != YH_OK) expect_failed("invoke failed", yh_group_last_error(g_));   // yolact.rs:163
}

FrameDetections YolactGroup::detections(int frame, bool want_masks) {
    FrameDetections out;
    out.dets.resize(static_cast<std::size_t>(max_dets_));
    if (want_masks) out.masks.resize(static_cast<std::size_t>(max_dets_) * hp_ * wp_);
    std::int32_t nd = 0;
    if (yh_group_read_detections(g_, frame, &nd, out.dets.data(), max_dets_, want_masks ? out.masks.data() : nullptr, out.masks.size()) != YH_OK)
        expect_failed("must data", yh_group_last_error(g_));                                                 // yolact.rs:173,:180
    out.dets.resize(static_cast<std::size_t>(nd));
    if (want_masks) out.masks.resize(static_cast<std::size_t>(nd) * hp_ * wp_);
    return out;
}

void YolactGroup::sync() {
    if (yh_group_sync(g_) != YH_OK) expect_failed("invoke failed", yh_group_last_error(g_));
}

}  // namespace tod
